// HBM-bound row-wise kernels of the X^2-VLM step (gfx950): LayerNorm fwd/bwd, layer-scale
// backward, column sums (bias gradients), dtype casts / weight transposes, patch extraction and
// token assembly, mean pooling.  One wave per row where rows are reduced (wavefront shuffle
// reductions, fp32 statistics), 16-byte accesses throughout.
//
// Reference sites: nn.LayerNorm in beit2.py:175,181,411 (eps 1e-6), xbert.py:214,422,506,796
// (eps 1e-12), xvlm.py:166 (eps 1e-5); gamma_1/gamma_2 layer scale beit2.py:206-207;
// PatchEmbed beit2.py:225-232; cls-token concat beit2.py:385-387; mean pooling beit2.py:409-416.
#include "x2_common.h"

// row r of a (B, period+1, D) token tensor with token 0 skipped  (period == 0: plain rows)
__device__ __forceinline__ long remap_row(int r, int period) { return period > 0 ? (long)r + r / period + 1 : r; }

// ---------------------------------------------------------------------------------- LayerNorm fwd
// x fp32 [rows][D] -> y_bf16 / y_f32 (either may be null), mean/rstd saved for the backward.
#define LN_MAXV 8   // float4 per lane: D <= 2048
// NV = float4 per lane actually used (ceil(D / 256)): keeps the register footprint (hence occupancy) tied to D
#define LN_DISPATCH(D, CALL) do { const int nv_ = ((D) + 255) / 256; \
  if (nv_ <= 1) { CALL(1); } else if (nv_ == 2) { CALL(2); } else if (nv_ == 3) { CALL(3); } else if (nv_ == 4) { CALL(4); } \
  else if (nv_ <= 6) { CALL(6); } else { CALL(8); } } while (0)
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bsh, bf16_t* yb, float* yf,
                                                            float* mean, float* rstd, int rows, int D, float eps, int period,
                                                            DropSpec drop_in_, const uint32_t* __restrict__ epoch) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const DropSpec drop = drop_at_epoch(drop_in_, epoch);
  const long gr = remap_row(row, period);
  const float* xr = x + gr * D;
  float4 v[NV];
  float s = 0.f;
  const int nv = D >> 2;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) { v[i] = *reinterpret_cast<const float4*>(xr + c * 4); s += v[i].x + v[i].y + v[i].z + v[i].w; }
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) { const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu; q += a * a + b * b + cc * cc + d * d; }
  }
  const float rs = rsqrtf(wave_sum(q) / D + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 ww = *reinterpret_cast<const float4*>(w + c * 4), bb = *reinterpret_cast<const float4*>(bsh + c * 4);
      float4 o{(v[i].x - mu) * rs * ww.x + bb.x, (v[i].y - mu) * rs * ww.y + bb.y, (v[i].z - mu) * rs * ww.z + bb.z, (v[i].w - mu) * rs * ww.w + bb.w};
      if (drop.thr16) {       // dropout on the LN output (BertEmbeddings, xbert.py:215)
        float dm[4];
        drop_mul4(drop, (uint32_t)gr * (uint32_t)D + (uint32_t)(c * 4), dm);
        o.x *= dm[0]; o.y *= dm[1]; o.z *= dm[2]; o.w *= dm[3];
      }
      if (yf) *reinterpret_cast<float4*>(yf + gr * D + c * 4) = o;
      if (yb) *reinterpret_cast<u32x2*>(yb + gr * D + c * 4) = u32x2{pack_bf16(o.x, o.y), pack_bf16(o.z, o.w)};
    }
  }
}

// Several rows per wave (RPW = 2 or 4), all their loads issued before the first reduction: a wave has RPW x NV 16-byte loads in
// flight instead of NV, a launch has 1 / RPW of the workgroups (a 12608-row LayerNorm was 3152 workgroups of ~1 us each: the
// dispatcher and the ramp-up / drain of a 10-17 us kernel, not HBM, set its time - 3.4 TB/s).  Same arithmetic per row, same
// summation order inside a row (bit-identical outputs to the one-row form).
template <int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bsh, bf16_t* yb, float* yf,
                                                                 float* mean, float* rstd, int rows, int D, float eps, int period,
                                                                 DropSpec drop_in_, const uint32_t* __restrict__ epoch) {
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW, lane = threadIdx.x & 63;
  if (row0 >= rows) return;
  const DropSpec drop = drop_at_epoch(drop_in_, epoch);
  const int nv = D >> 2;
  float4 v[RPW][NV];
  long gr[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int row = min(row0 + r, rows - 1);          // rows past the end re-read the last row (results discarded)
    gr[r] = remap_row(row, period);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      v[r][i] = c < nv ? *reinterpret_cast<const float4*>(x + gr[r] * D + c * 4) : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float4 ww[NV], bb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    ww[i] = c < nv ? *reinterpret_cast<const float4*>(w + c * 4) : float4{0.f, 0.f, 0.f, 0.f};
    bb[i] = c < nv ? *reinterpret_cast<const float4*>(bsh + c * 4) : float4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (lane + i * 64 < nv) s += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
    const float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (lane + i * 64 < nv) { const float a = v[r][i].x - mu, b = v[r][i].y - mu, cc = v[r][i].z - mu, d = v[r][i].w - mu; q += a * a + b * b + cc * cc + d * d; }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    if (row0 + r >= rows) continue;
    if (lane == 0) { mean[row0 + r] = mu; rstd[row0 + r] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float4 o{(v[r][i].x - mu) * rs * ww[i].x + bb[i].x, (v[r][i].y - mu) * rs * ww[i].y + bb[i].y,
                 (v[r][i].z - mu) * rs * ww[i].z + bb[i].z, (v[r][i].w - mu) * rs * ww[i].w + bb[i].w};
        if (drop.thr16) {       // dropout on the LN output (BertEmbeddings, xbert.py:215)
          float dm[4];
          drop_mul4(drop, (uint32_t)gr[r] * (uint32_t)D + (uint32_t)(c * 4), dm);
          o.x *= dm[0]; o.y *= dm[1]; o.z *= dm[2]; o.w *= dm[3];
        }
        if (yf) *reinterpret_cast<float4*>(yf + gr[r] * D + c * 4) = o;
        if (yb) *reinterpret_cast<u32x2*>(yb + gr[r] * D + c * 4) = u32x2{pack_bf16(o.x, o.y), pack_bf16(o.z, o.w)};
      }
    }
  }
}

extern "C" int x2_tune_get(int key);
extern "C" int x2_layernorm_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32, float* mean,
                                float* rstd, int rows, int D, float eps, int period, unsigned drop_thr16, unsigned drop_seed,
                                float drop_scale, const unsigned* drop_epoch, void* stream) {
  X2_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAXV, "x2_layernorm_fwd: rows=%d D=%d (D%%4==0, D<=2048)", rows, D);
#define X2_LNF(NV) hipLaunchKernelGGL(layernorm_fwd_kernel<NV>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, b, \
                     (bf16_t*)y_bf16, y_f32, mean, rstd, rows, D, eps, period, DropSpec{drop_thr16, drop_seed, drop_scale}, drop_epoch)
#define X2_LNF_R(NV, RPW) hipLaunchKernelGGL((layernorm_fwd_rows_kernel<NV, RPW>), dim3((rows + 4 * RPW - 1) / (4 * RPW)), dim3(256), 0, (hipStream_t)stream, \
                     x, w, b, (bf16_t*)y_bf16, y_f32, mean, rstd, rows, D, eps, period, DropSpec{drop_thr16, drop_seed, drop_scale}, drop_epoch)
  // rows per wave: x2_tune(13, v): 0 = automatic, 1 = one (the round-1 form), 2, 4.  Automatic: 2 when the launch still has >= 4
  // workgroups per CU that way, else 1 (short launches keep every CU busy)
  const int knob = x2_tune_get(13), nvv = (D + 255) / 256;
  int rpw = knob == 1 || knob == 2 || knob == 4 ? knob : (rows >= 8 * 4 * 256 ? 2 : 1);
  if (nvv > 4) rpw = 1;                               // D > 1024: the one-row form's register footprint is already the limit
  if (rpw == 4 && nvv <= 3) { if (nvv <= 1) X2_LNF_R(1, 4); else if (nvv == 2) X2_LNF_R(2, 4); else X2_LNF_R(3, 4); }
  else if (rpw >= 2) { if (nvv <= 1) X2_LNF_R(1, 2); else if (nvv == 2) X2_LNF_R(2, 2); else if (nvv == 3) X2_LNF_R(3, 2); else X2_LNF_R(4, 2); }
  else LN_DISPATCH(D, X2_LNF);
  return x2_check_launch("x2_layernorm_fwd");
}

// ---------------------------------------------------------------------------------- partial-sum reducer
// Column reductions over many rows are done in two deterministic stages instead of fp32 atomics (measured
// on MI355X: ~43 G atomic-adds/s, i.e. 1.8 M atomics of a 12608-row LayerNorm backward cost more than its
// HBM traffic): stage 1 writes per-workgroup partial rows part[blk][k][width], this kernel adds them up:
// out_k[c] += sum_blk part[blk][k][c].
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nblk, int nk, int width,
                                                              float* o0, float* o1, float* o2, float* o3 = nullptr) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), k = blockIdx.y, sl = threadIdx.x >> 6;
  __shared__ float red[3][64];
  float s = 0.f;
  if (c < width) {
#pragma unroll 4
    for (int b = sl; b < nblk; b += 4) s += part[((long)b * nk + k) * width + c];
  }
  if (sl > 0) red[sl - 1][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && c < width) {
    float* o = k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : o3));
    if (o) o[c] += s + red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x];
  }
}
static void launch_reduce(const float* part, int nblk, int nk, int width, float* o0, float* o1, float* o2, hipStream_t st, float* o3 = nullptr) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((width + 63) / 64, nk), dim3(256), 0, st, part, nblk, nk, width, o0, o1, o2, o3);
}
// Stage 2 on its own: the parameter-gradient sums are not on the critical path of the backward, so the host may
// run this on another stream (stage 1 launched with defer = 1 and a private workspace).
extern "C" int x2_reduce_partials(const float* part, int nblk, int nk, int width, float* o0, float* o1, float* o2, void* stream) {
  X2_REQUIRE(part && nblk > 0 && nk >= 1 && nk <= 3 && width > 0, "x2_reduce_partials: nblk=%d nk=%d width=%d", nblk, nk, width);
  launch_reduce(part, nblk, nk, width, o0, o1, o2, (hipStream_t)stream);
  return x2_check_launch("x2_reduce_partials");
}

// Several stage-2 reductions in one launch (all parameter-gradient sums of one layer's backward): the reductions
// are tiny (a few dozen workgroups each), so one launch per reduction is pure launch latency on both sides.
// desc rows of 8 int64: {part, nblk, nk, width, o0, o1, o2, o3}.
#define RPM_MAX 16
struct ReduceDesc { const float* part; float* o[4]; int nblk, nk, width, blk0; };
struct ReduceGroup { ReduceDesc d[RPM_MAX]; int count; };
// Thread mapping: 4 column groups of 4 floats (one 16-column tile = one 64-byte segment per partial row and workgroup) x 64 row
// slices; a thread walks the partial rows b = slice, slice + 64, ... with 16-byte loads, four in flight; the 64 slices are
// added in a fixed order through LDS.  The reduction is a latency chain, not a bandwidth problem (a few MB per launch): what
// counts is how few dependent load batches a thread issues and how many CUs take part - with 64-column tiles and 16 slices a
// 788-row LayerNorm workspace was 12 batches deep on 36 workgroups (11.6 us for 7.3 MB; probes/bench_rowwise.py), this form is 3-4
// batches on 144.  (The step's own launches carry 8-16 workspaces, 48 MB: 33.7 -> 32.2 us, 1.5 TB/s either way.)
#define RPM_COLS 16
__global__ __launch_bounds__(256) void reduce_partials_multi_kernel(ReduceGroup g) {
  int e = 0;
#pragma unroll
  for (int i = 1; i < RPM_MAX; ++i) if (i < g.count && (int)blockIdx.x >= g.d[i].blk0) e = i;
  const ReduceDesc d = g.d[e];
  const int local = blockIdx.x - d.blk0, ctiles = (d.width + RPM_COLS - 1) / RPM_COLS;
  const int k = local / ctiles, cg = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const int c = (local % ctiles) * RPM_COLS + cg * 4;
  __shared__ float4 red[63][4];
  float4 s = float4{0.f, 0.f, 0.f, 0.f};
  float* o = d.o[k];
  const bool vec = (d.width & 3) == 0;                  // every row 16-byte aligned (the workspaces are)
  if (o && c < d.width) {
    const float* base = d.part + (size_t)k * d.width + c;
    const size_t rs = (size_t)d.nk * d.width;
    if (vec) {
      int b = sl;
      for (; b + 192 < d.nblk; b += 256) {
        const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)b * rs), v1 = *reinterpret_cast<const float4*>(base + (size_t)(b + 64) * rs),
                     v2 = *reinterpret_cast<const float4*>(base + (size_t)(b + 128) * rs), v3 = *reinterpret_cast<const float4*>(base + (size_t)(b + 192) * rs);
        s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
        s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
      }
      for (; b < d.nblk; b += 64) {
        const float4 v = *reinterpret_cast<const float4*>(base + (size_t)b * rs);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    } else {
      for (int b = sl; b < d.nblk; b += 64) {
        const float* r = base + (size_t)b * rs;
        s.x += r[0]; if (c + 1 < d.width) s.y += r[1]; if (c + 2 < d.width) s.z += r[2]; if (c + 3 < d.width) s.w += r[3];
      }
    }
  }
  if (sl > 0) red[sl - 1][cg] = s;
  __syncthreads();
  if (sl == 0 && o && c < d.width) {
#pragma unroll 9
    for (int i = 0; i < 63; ++i) { const float4 v = red[i][cg]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    o[c] += s.x;
    if (c + 1 < d.width) o[c + 1] += s.y;
    if (c + 2 < d.width) o[c + 2] += s.z;
    if (c + 3 < d.width) o[c + 3] += s.w;
  }
}
extern "C" int x2_reduce_partials_multi(const int64_t* desc, int count, void* stream) {
  X2_REQUIRE(desc && count >= 1, "x2_reduce_partials_multi: count=%d", count);
  for (int i0 = 0; i0 < count; i0 += RPM_MAX) {
    ReduceGroup g; g.count = count - i0 < RPM_MAX ? count - i0 : RPM_MAX;
    int blocks = 0;
    for (int i = 0; i < g.count; ++i) {
      const int64_t* q = desc + (size_t)(i0 + i) * 8;
      ReduceDesc& d = g.d[i];
      d.part = (const float*)q[0]; d.nblk = (int)q[1]; d.nk = (int)q[2]; d.width = (int)q[3];
      d.o[0] = (float*)q[4]; d.o[1] = (float*)q[5]; d.o[2] = (float*)q[6]; d.o[3] = (float*)q[7];
      X2_REQUIRE(d.part && d.nblk > 0 && d.nk >= 1 && d.nk <= 4 && d.width > 0, "x2_reduce_partials_multi[%d]: nblk=%d nk=%d width=%d",
                 i0 + i, d.nblk, d.nk, d.width);
      d.blk0 = blocks;
      blocks += ((d.width + RPM_COLS - 1) / RPM_COLS) * d.nk;
    }
    hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  }
  return x2_check_launch("x2_reduce_partials_multi");
}

// ---------------------------------------------------------------------------------- LayerNorm bwd
// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w ; dw += sum dy*xhat ; db += sum dy ;
// optionally dcol += sum_rows dx (the bias gradient of the linear layer that produced the LN input).
// A workgroup (4 waves) walks LNB_ROWS rows; each wave keeps its dw/db/dcol partials in registers, the
// four waves are combined through LDS and written as one partial row per workgroup (ws[blk][3][D]);
// reduce_partials_kernel then adds the partial rows into dw/db/dcol.
#define LNB_ROWS 16
// DYB: the incoming gradient dy is bf16 (the input-gradient GEMM of a pre-LN block writes bf16, as the reference's apex-O1
// linears hand fp16 gradients to their LayerNorm: 2 bytes instead of 4 written by the GEMM and read here)
// (A variant that also did the layer-scale backward its output feeds - x2_layernorm_bwd_layerscale, rounds 3-4 - saved a 38.7 MB read per
// vision block but needed 156 VGPRs (3 waves per SIMD): 25.16 vs 24.92 ms per base step; with LDS atomics at 128 VGPRs 25.22 vs 23.58;
// forced to 128 VGPRs with 29 spilled dwords 24.05-24.17 vs 23.59-23.62 (profiles/r05j_fused_ln_layerscale_ab.txt).  Removed.)
// MODE 0: dy fp32; 1: dy bf16; 2: dy bf16 and the by-products refer to the FINAL output times a per-row factor: the bf16 copy is
// post_rs[row] * (dx + dres) and the third partial row its column sums - what the layer scale BELOW this LayerNorm needs of
// its incoming gradient (x2_layerscale_finish; post_rs = DropPath factor of that branch or null).
template <int NV, int MODE>
__global__ __launch_bounds__(256, NV <= 3 ? 4 : 1) void layernorm_bwd_kernel(const void* __restrict__ dy_, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ w, const float* dres, float* dx, bf16_t* dxb,
                                                            float* ws, int rows, int D, int period,
                                                            DropSpec din_, DropSpec dout_, const uint32_t* __restrict__ epoch,
                                                            const float* __restrict__ post_rs) {
  constexpr int NSET = 3;
  constexpr bool DYB = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) float red[];      // [NSET][3 waves][D]
  const DropSpec din = drop_at_epoch(din_, epoch), dout = drop_at_epoch(dout_, epoch);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = D >> 2;
  float4 ww[NV], aw[NV], ab[NV], ac[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    ww[i] = c < nv ? *reinterpret_cast<const float4*>(w + c * 4) : float4{0.f, 0.f, 0.f, 0.f};
    aw[i] = float4{0.f, 0.f, 0.f, 0.f}; ab[i] = float4{0.f, 0.f, 0.f, 0.f}; ac[i] = float4{0.f, 0.f, 0.f, 0.f};
  }
  const int r0 = blockIdx.x * LNB_ROWS, r1 = min(rows, r0 + LNB_ROWS);
  for (int row = r0 + wv; row < r1; row += 4) {
    const long gr = remap_row(row, period);
    const float mu = mean[row], rs = rstd[row];
    float4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float4 d;
        if constexpr (DYB) {
          const u32x2 raw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dy_) + gr * D + c * 4);
          d = float4{bf_lo(raw.x), bf_hi(raw.x), bf_lo(raw.y), bf_hi(raw.y)};
        } else {
          d = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + gr * D + c * 4);
        }
        const float4 xv = *reinterpret_cast<const float4*>(x + gr * D + c * 4);
        if (din.thr16) {      // the forward dropped the LN OUTPUT: mask the incoming gradient the same way
          float dm[4];
          drop_mul4(din, (uint32_t)gr * (uint32_t)D + (uint32_t)(c * 4), dm);
          d.x *= dm[0]; d.y *= dm[1]; d.z *= dm[2]; d.w *= dm[3];
        }
        xh[i] = float4{(xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs};
        g[i] = float4{d.x * ww[i].x, d.y * ww[i].y, d.z * ww[i].z, d.w * ww[i].w};
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
        aw[i].x += d.x * xh[i].x; aw[i].y += d.y * xh[i].y; aw[i].z += d.z * xh[i].z; aw[i].w += d.w * xh[i].w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
      }
    }
    const float m1 = wave_sum(s1) / D, m2 = wave_sum(s2) / D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float4 o{rs * (g[i].x - m1 - xh[i].x * m2), rs * (g[i].y - m1 - xh[i].y * m2), rs * (g[i].z - m1 - xh[i].z * m2), rs * (g[i].w - m1 - xh[i].w * m2)};
        // the producing linear's output was dropped before the residual add: its gradient (bf16 copy, bias sums)
        // carries that mask, the residual branch (dx) does not
        float4 om = o;
        if (dout.thr16) {
          float dm[4];
          drop_mul4(dout, (uint32_t)gr * (uint32_t)D + (uint32_t)(c * 4), dm);
          om.x *= dm[0]; om.y *= dm[1]; om.z *= dm[2]; om.w *= dm[3];
        }
        if (dres) { const float4 rr = *reinterpret_cast<const float4*>(dres + gr * D + c * 4); o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
        if (dx) *reinterpret_cast<float4*>(dx + gr * D + c * 4) = o;
        if constexpr (MODE == 2) { om = o; if (post_rs) { const float f = post_rs[gr]; om.x *= f; om.y *= f; om.z *= f; om.w *= f; } }
        ac[i].x += om.x; ac[i].y += om.y; ac[i].z += om.z; ac[i].w += om.w;
        if (dxb) *reinterpret_cast<u32x2*>(dxb + gr * D + c * 4) = u32x2{pack_bf16(om.x, om.y), pack_bf16(om.z, om.w)};
      }
    }
  }
  // combine the 4 waves: waves 1..3 park their partials in LDS, wave 0 adds them and issues the atomics
  if (wv > 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        *reinterpret_cast<float4*>(red + ((0 * 3 + wv - 1) * D) + c * 4) = aw[i];
        *reinterpret_cast<float4*>(red + ((1 * 3 + wv - 1) * D) + c * 4) = ab[i];
        *reinterpret_cast<float4*>(red + ((2 * 3 + wv - 1) * D) + c * 4) = ac[i];
      }
    }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float4 a = *reinterpret_cast<const float4*>(red + ((0 * 3 + k) * D) + c * 4), b = *reinterpret_cast<const float4*>(red + ((1 * 3 + k) * D) + c * 4),
                       cc = *reinterpret_cast<const float4*>(red + ((2 * 3 + k) * D) + c * 4);
          aw[i].x += a.x; aw[i].y += a.y; aw[i].z += a.z; aw[i].w += a.w;
          ab[i].x += b.x; ab[i].y += b.y; ab[i].z += b.z; ab[i].w += b.w;
          ac[i].x += cc.x; ac[i].y += cc.y; ac[i].z += cc.z; ac[i].w += cc.w;
        }
        float* wp = ws + (long)blockIdx.x * NSET * D + c * 4;
        *reinterpret_cast<float4*>(wp) = aw[i];
        *reinterpret_cast<float4*>(wp + D) = ab[i];
        *reinterpret_cast<float4*>(wp + 2 * D) = ac[i];
      }
    }
  }
}

extern "C" int x2_layernorm_bwd(const void* dy, int dy_is_bf16, const float* x, const float* mean, const float* rstd, const float* w,
                                const float* dres, float* dx, void* dx_bf16, float* dw, float* db, float* dcol, int rows, int D,
                                int period, unsigned in_thr16, unsigned in_seed, float in_scale, unsigned out_thr16,
                                unsigned out_seed, float out_scale, const unsigned* drop_epoch, float* ws, int defer, int post,
                                const float* post_rowscale, void* stream) {
  X2_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 256 * LN_MAXV, "x2_layernorm_bwd: rows=%d D=%d", rows, D);
  X2_REQUIRE(dw && db && ws, "x2_layernorm_bwd: dw/db and the workspace ws[ceil(rows/%d)*3*D] are required", LNB_ROWS);
  X2_REQUIRE(post || !(dcol && dres), "x2_layernorm_bwd: dcol sums the LN-input gradient, which excludes dres (post = 0)");
  X2_REQUIRE(!(out_thr16 && dres), "x2_layernorm_bwd: an output mask applies to the bf16 copy, which excludes dres");
  X2_REQUIRE(!post || (dy_is_bf16 && dx_bf16 && dcol && !out_thr16 && period == 0),
             "x2_layernorm_bwd: post = 1 (by-products of the final output) takes a bf16 dy, dx_bf16 and dcol, no output mask, period 0");
#define X2_LNB_T(NV, B) hipLaunchKernelGGL((layernorm_bwd_kernel<NV, B>), dim3((rows + LNB_ROWS - 1) / LNB_ROWS), dim3(256), 9 * D * sizeof(float), \
                     (hipStream_t)stream, dy, x, mean, rstd, w, dres, dx, (bf16_t*)dx_bf16, ws, rows, D, period,                \
                     DropSpec{in_thr16, in_seed, in_scale}, DropSpec{out_thr16, out_seed, out_scale}, drop_epoch, post_rowscale)
#define X2_LNB(NV) do { if (post) X2_LNB_T(NV, 2); else if (dy_is_bf16) X2_LNB_T(NV, 1); else X2_LNB_T(NV, 0); } while (0)
  LN_DISPATCH(D, X2_LNB);
  if (!defer) launch_reduce(ws, (rows + LNB_ROWS - 1) / LNB_ROWS, 3, D, dw, db, dcol, (hipStream_t)stream);
  return x2_check_launch("x2_layernorm_bwd");
}

// ---------------------------------------------------------------------------------- column sums
// out[n] += sum_m Y[m][n]  (bias gradients).  Workgroup = 512 columns x CS_ROWS rows: 64 lanes x 16 B
// across, 4 waves down; waves combined through LDS into one partial row ws[rowblk][N], then reduced.
#define CS_ROWS 64
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ y, float* ws, int M, int N, int ld) {
  __shared__ float red[3][512];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = blockIdx.y * 512 + tx * 8;
  const int r0 = blockIdx.x * CS_ROWS, r1 = min(M, r0 + CS_ROWS);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < N) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 4) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(y + (long)r * ld + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += bf_lo(v[e]); acc[2 * e + 1] += bf_hi(v[e]); }
    }
  }
  if (ty > 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty - 1][tx * 8 + e] = acc[e];
  }
  __syncthreads();
  if (ty == 0 && c0 < N) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = acc[e] + red[0][tx * 8 + e] + red[1][tx * 8 + e] + red[2][tx * 8 + e];
      if (c0 + e < N) ws[(long)blockIdx.x * N + c0 + e] = v;
    }
  }
}
extern "C" int x2_colsum_bf16(const void* y, float* out, int M, int N, int ld, float* ws, int defer, void* stream) {
  X2_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0, "x2_colsum_bf16: M=%d N=%d ld=%d (N, ld multiples of 8)", M, N, ld);
  X2_REQUIRE(ws, "x2_colsum_bf16: workspace ws[ceil(M/%d)*N] required", CS_ROWS);
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((M + CS_ROWS - 1) / CS_ROWS, (N + 511) / 512), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y, ws, M, N, ld);
  if (!defer) launch_reduce(ws, (M + CS_ROWS - 1) / CS_ROWS, 1, N, out, nullptr, nullptr, (hipStream_t)stream);
  return x2_check_launch("x2_colsum_bf16");
}

// ---------------------------------------------------------------------------------- layer-scale backward
// forward was  x_out = x_in + r[m] * gamma * u,  u = A . W^T + b  (r = DropPath row factor or 1).  With dX' = r[m] * dX (fp32):
//   dA     = (gamma * dX') . W        = dX' . (diag(gamma) W)      the GEMM reads a weight copy with gamma folded in (cast_transpose_multi)
//   dW     = (gamma * dX')^T . A      = diag(gamma) G,  G = dX'^T . A   the weight-gradient GEMM runs on dX' itself
//   db     = gamma * colsum(dX')
//   dgamma = sum_m dX' * u            = rowdot(G, W) + b * colsum(dX')   (u = A . W^T + b substituted: no pass over activations)
// so the backward needs from the [M, D] tensors only a bf16 copy of dX' and its column sums - both by-products of the
// LayerNorm backward that produces dX (MODE 2 there) - and x2_layerscale_finish, a pass over the [D, K] weight gradient.
// Against the kernel this replaces (dU = gamma * dX as a separate pass over dX and the saved u): 77 MB less traffic per layer scale
// in the backward and no 19 MB side output `u` from the forward GEMM (X2VLM-base, 12608 x 768).
//
// x2_rowscale_cast_colsum: the same two by-products from a stand-alone pass, for a dX that no LayerNorm backward of this stage
// produced (top of a tower / of a chunk of blocks).  Workgroup = 256 columns x LS_ROWS rows, partial rows ws[blk][D].
#define LS_ROWS 32
__global__ __launch_bounds__(256) void rowscale_cast_colsum_kernel(const float* __restrict__ dx, const float* __restrict__ rowscale,
                                                                   bf16_t* dxb, float* ws, int M, int D) {
  __shared__ float red[3][256];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c0 = blockIdx.y * 256 + tx * 4;
  const int r0 = blockIdx.x * LS_ROWS, r1 = min(M, r0 + LS_ROWS);
  float ab[4] = {0.f, 0.f, 0.f, 0.f};
  if (c0 < D) {
#pragma unroll 4
    for (int r = r0 + ty; r < r1; r += 4) {
      float4 d = *reinterpret_cast<const float4*>(dx + (long)r * D + c0);
      if (rowscale) { const float rs_ = rowscale[r]; d.x *= rs_; d.y *= rs_; d.z *= rs_; d.w *= rs_; }   // DropPath keep/(1-p)
      *reinterpret_cast<u32x2*>(dxb + (long)r * D + c0) = u32x2{pack_bf16(d.x, d.y), pack_bf16(d.z, d.w)};
      ab[0] += d.x; ab[1] += d.y; ab[2] += d.z; ab[3] += d.w;
    }
  }
  if (ty > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ty - 1][tx * 4 + e] = ab[e];
  }
  __syncthreads();
  if (ty == 0 && c0 < D) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ws[(long)blockIdx.x * D + c0 + e] = ab[e] + red[0][tx * 4 + e] + red[1][tx * 4 + e] + red[2][tx * 4 + e];
  }
}
extern "C" int x2_rowscale_cast_colsum(const float* dx, const float* rowscale, void* dx_bf16, float* colsum, int M, int D, float* ws,
                                       int defer, void* stream) {
  X2_REQUIRE(dx && dx_bf16 && colsum && M > 0 && D > 0 && D % 4 == 0 && ws,
             "x2_rowscale_cast_colsum: M=%d D=%d (workspace ws[ceil(M/%d)*D] required)", M, D, LS_ROWS);
  hipLaunchKernelGGL(rowscale_cast_colsum_kernel, dim3((M + LS_ROWS - 1) / LS_ROWS, (D + 255) / 256), dim3(256), 0, (hipStream_t)stream, dx,
                     rowscale, (bf16_t*)dx_bf16, ws, M, D);
  if (!defer) launch_reduce(ws, (M + LS_ROWS - 1) / LS_ROWS, 1, D, colsum, nullptr, nullptr, (hipStream_t)stream);
  return x2_check_launch("x2_rowscale_cast_colsum");
}

// One wave per row n of a [N, K] weight gradient G = dX'^T . A (as the TN GEMM left it):
//   dgamma[n] += rowdot(G[n], W[n]) + bias[n] * cs[n];   dbias[n] += gamma[n] * cs[n];   G[n][:] *= gamma[n]
// desc rows of 9 int64: {G, W, bias or 0, gamma, cs, dgamma, dbias or 0, N, K}; W the fp32 master weight; K % 4 == 0.
#define LSF_MAX 8
struct LsfDesc { float* G; const float* W; const float* bias; const float* gamma; const float* cs; float* dgamma; float* dbias; int N, K, blk0; };
struct LsfGroup { LsfDesc d[LSF_MAX]; int count; };
__global__ __launch_bounds__(256) void layerscale_finish_kernel(LsfGroup g) {
  int e = 0;
#pragma unroll
  for (int i = 1; i < LSF_MAX; ++i) if (i < g.count && (int)blockIdx.x >= g.d[i].blk0) e = i;
  const LsfDesc d = g.d[e];
  const int n = (blockIdx.x - d.blk0) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= d.N) return;
  float* gr = d.G + (size_t)n * d.K;
  const float* wr = d.W + (size_t)n * d.K;
  const float gm = d.gamma[n];
  float dot = 0.f;
  for (int k = lane * 4; k < d.K; k += 256) {
    float4 gv = *reinterpret_cast<const float4*>(gr + k);
    const float4 wv = *reinterpret_cast<const float4*>(wr + k);
    dot += gv.x * wv.x + gv.y * wv.y + gv.z * wv.z + gv.w * wv.w;
    gv.x *= gm; gv.y *= gm; gv.z *= gm; gv.w *= gm;
    *reinterpret_cast<float4*>(gr + k) = gv;
  }
  dot = wave_sum(dot);
  if (lane == 0) {
    const float c = d.cs[n];
    d.dgamma[n] += dot + (d.bias ? d.bias[n] * c : 0.f);
    if (d.dbias) d.dbias[n] += gm * c;
  }
}
extern "C" int x2_layerscale_finish(const int64_t* desc, int count, void* stream) {
  X2_REQUIRE(desc && count >= 1, "x2_layerscale_finish: count=%d", count);
  for (int i0 = 0; i0 < count; i0 += LSF_MAX) {
    LsfGroup g; g.count = count - i0 < LSF_MAX ? count - i0 : LSF_MAX;
    int blocks = 0;
    for (int i = 0; i < g.count; ++i) {
      const int64_t* q = desc + (size_t)(i0 + i) * 9;
      LsfDesc& d = g.d[i];
      d.G = (float*)q[0]; d.W = (const float*)q[1]; d.bias = (const float*)q[2]; d.gamma = (const float*)q[3]; d.cs = (const float*)q[4];
      d.dgamma = (float*)q[5]; d.dbias = (float*)q[6]; d.N = (int)q[7]; d.K = (int)q[8];
      X2_REQUIRE(d.G && d.W && d.gamma && d.cs && d.dgamma && d.N > 0 && d.K > 0 && d.K % 4 == 0, "x2_layerscale_finish[%d]: N=%d K=%d", i0 + i,
                 d.N, d.K);
      d.blk0 = blocks;
      blocks += (d.N + 3) / 4;
    }
    hipLaunchKernelGGL(layerscale_finish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  }
  return x2_check_launch("x2_layerscale_finish");
}

// ---------------------------------------------------------------------------------- casts
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n4) {
  long i = blockIdx.x * 256L + threadIdx.x;
  const long st = (long)gridDim.x * 256;
  for (; i < n4; i += st) {
    const float4 v = *reinterpret_cast<const float4*>(s + i * 4);
    *reinterpret_cast<u32x2*>(d + i * 4) = u32x2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
  }
}
extern "C" int x2_cast_bf16(const float* src, void* dst, long n, void* stream) {
  X2_REQUIRE(n > 0 && n % 4 == 0, "x2_cast_bf16: n=%ld must be a positive multiple of 4", n);
  const long n4 = n / 4;
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n4);
  return x2_check_launch("x2_cast_bf16");
}

// fp32 [R][C] -> bf16 [R][C] (optional) and bf16 transposed [C][ldt] (ldt >= R; columns R..ldt-1 left untouched).
// 64x64 tiles through LDS; 16-byte loads, 8-byte stores on both sides when R, C, ldt are multiples of 4.
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ s, bf16_t* d, bf16_t* dT, int R, int C, int ldt) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16 threads, 4 columns each
  const bool vec = ((C | ldt | R) & 3) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
    float4 v{0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      if (vec && c + 3 < C) {
        v = *reinterpret_cast<const float4*>(s + (long)r * C + c);
        if (d) *reinterpret_cast<u32x2*>(d + (long)r * C + c) = u32x2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
      } else {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < 4; ++e) if (c + e < C) { t[e] = s[(long)r * C + c + e]; if (d) d[(long)r * C + c + e] = f2bf(t[e]); }
        v = float4{t[0], t[1], t[2], t[3]};
      }
    }
    tile[ty + 16 * i][tx * 4 + 0] = v.x; tile[ty + 16 * i][tx * 4 + 1] = v.y;
    tile[ty + 16 * i][tx * 4 + 2] = v.z; tile[ty + 16 * i][tx * 4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 16 * i, r = r0 + tx * 4;               // output row = source column
    if (c >= C) continue;
    const float a0 = tile[tx * 4 + 0][ty + 16 * i], a1 = tile[tx * 4 + 1][ty + 16 * i],
                a2 = tile[tx * 4 + 2][ty + 16 * i], a3 = tile[tx * 4 + 3][ty + 16 * i];
    if (vec && r + 3 < R) {
      *reinterpret_cast<u32x2*>(dT + (long)c * ldt + r) = u32x2{pack_bf16(a0, a1), pack_bf16(a2, a3)};
    } else {
      const float t[4] = {a0, a1, a2, a3};
      for (int e = 0; e < 4; ++e) if (r + e < R) dT[(long)c * ldt + r + e] = f2bf(t[e]);
    }
  }
}
extern "C" int x2_cast_transpose_bf16(const float* src, void* dst, void* dstT, int R, int C, int ldt, void* stream) {
  X2_REQUIRE(R > 0 && C > 0 && ldt >= R && dstT, "x2_cast_transpose_bf16: R=%d C=%d ldt=%d", R, C, ldt);
  hipLaunchKernelGGL(cast_transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                     (bf16_t*)dstT, R, C, ldt);
  return x2_check_launch("x2_cast_transpose_bf16");
}

// The bf16 copies of MANY fp32 weights in one launch (all linears of a tower: once per optimizer step).  Entry i
// casts src_i [R_i][C_i] into rows roff_i.. of a bf16 matrix dst [*][C_i] and into columns roff_i.. of the transposed
// bf16 matrix dstT [C_i][ldt_i], so weights that are used stacked (q/k/v) need no concatenation pass.
// desc rows of 8 int64: {src, dst, dstT, R, C, ldt, roff, tscale}; R, C, ldt, roff multiples of 4.  tscale (fp32 [R] or 0): the
// TRANSPOSED copy holds tscale[r] * src[r][c] - a BEiT layer scale folded into the weight its input-gradient GEMM reads
// (dA = (gamma * dX) . W = dX . (diag(gamma) W): the backward then never forms gamma * dX; x2_layerscale_finish).
#define CTM_MAX 48
struct CastDesc { const float* s; bf16_t* d; bf16_t* dT; const float* sc; int R, C, ldt, blk0; };
struct CastGroup { CastDesc e[CTM_MAX]; int count; };
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(CastGroup g) {
  __shared__ float tile[64][65];
  int lo = 0, hi = g.count - 1;                    // last entry whose first block is <= blockIdx.x
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)blockIdx.x >= g.e[mid].blk0) lo = mid; else hi = mid - 1; }
  const CastDesc d = g.e[lo];
  const int local = blockIdx.x - d.blk0, ct = (d.C + 63) / 64;
  const int r0 = (local / ct) * 64, c0 = (local % ct) * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 16 * i, c = c0 + tx * 4;
    float4 v{0.f, 0.f, 0.f, 0.f};
    if (r < d.R && c < d.C) {
      v = *reinterpret_cast<const float4*>(d.s + (long)r * d.C + c);
      *reinterpret_cast<u32x2*>(d.d + (long)r * d.C + c) = u32x2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
    }
    if (d.sc && r < d.R) { const float f = d.sc[r]; v.x *= f; v.y *= f; v.z *= f; v.w *= f; }
    tile[ty + 16 * i][tx * 4 + 0] = v.x; tile[ty + 16 * i][tx * 4 + 1] = v.y;
    tile[ty + 16 * i][tx * 4 + 2] = v.z; tile[ty + 16 * i][tx * 4 + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 16 * i, r = r0 + tx * 4;
    if (c >= d.C || r >= d.R) continue;
    *reinterpret_cast<u32x2*>(d.dT + (long)c * d.ldt + r) =
        u32x2{pack_bf16(tile[tx * 4 + 0][ty + 16 * i], tile[tx * 4 + 1][ty + 16 * i]),
              pack_bf16(tile[tx * 4 + 2][ty + 16 * i], tile[tx * 4 + 3][ty + 16 * i])};
  }
}
extern "C" int x2_cast_transpose_multi(const int64_t* desc, int count, void* stream) {
  X2_REQUIRE(desc && count >= 1, "x2_cast_transpose_multi: count=%d", count);
  for (int i0 = 0; i0 < count; i0 += CTM_MAX) {
    CastGroup g; g.count = count - i0 < CTM_MAX ? count - i0 : CTM_MAX;
    int blocks = 0;
    for (int i = 0; i < g.count; ++i) {
      const int64_t* q = desc + (size_t)(i0 + i) * 8;
      CastDesc& d = g.e[i];
      d.R = (int)q[3]; d.C = (int)q[4]; d.ldt = (int)q[5];
      const int roff = (int)q[6];
      X2_REQUIRE(q[0] && q[1] && q[2] && d.R > 0 && d.C > 0 && roff >= 0 && d.ldt >= roff + d.R && ((d.R | d.C | d.ldt | roff) & 3) == 0,
                 "x2_cast_transpose_multi[%d]: R=%d C=%d ldt=%d roff=%d (multiples of 4 required)", i0 + i, d.R, d.C, d.ldt, roff);
      d.s = (const float*)q[0];
      d.d = (bf16_t*)q[1] + (size_t)roff * d.C;
      d.dT = (bf16_t*)q[2] + roff;
      d.sc = (const float*)q[7];
      d.blk0 = blocks;
      blocks += ((d.R + 63) / 64) * ((d.C + 63) / 64);
    }
    hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  }
  return x2_check_launch("x2_cast_transpose_multi");
}

// Many small fp32 vectors packed into one buffer in one launch (the stacked q/k/v biases of every layer of a tower, with
// zero segments where the reference has no bias: BEiT's k).  desc rows of 3 int64: {src or 0 (= zeros), dst, n}.
#define CPM_MAX 96
struct CopyDesc { const float* s; float* d; int n; int blk0; };
struct CopyGroup { CopyDesc e[CPM_MAX]; int count; };
__global__ __launch_bounds__(256) void copy_f32_multi_kernel(CopyGroup g) {
  int lo = 0, hi = g.count - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((int)blockIdx.x >= g.e[mid].blk0) lo = mid; else hi = mid - 1; }
  const CopyDesc d = g.e[lo];
  const int i = (blockIdx.x - d.blk0) * 256 + threadIdx.x;
  if (i < d.n) d.d[i] = d.s ? d.s[i] : 0.f;
}
extern "C" int x2_copy_f32_multi(const int64_t* desc, int count, void* stream) {
  X2_REQUIRE(desc && count >= 1, "x2_copy_f32_multi: count=%d", count);
  for (int i0 = 0; i0 < count; i0 += CPM_MAX) {
    CopyGroup g; g.count = count - i0 < CPM_MAX ? count - i0 : CPM_MAX;
    int blocks = 0;
    for (int i = 0; i < g.count; ++i) {
      const int64_t* q = desc + (size_t)(i0 + i) * 3;
      X2_REQUIRE(q[1] && q[2] > 0, "x2_copy_f32_multi[%d]: dst=%p n=%ld", i0 + i, (void*)q[1], (long)q[2]);
      g.e[i] = CopyDesc{(const float*)q[0], (float*)q[1], (int)q[2], blocks};
      blocks += ((int)q[2] + 255) / 256;
    }
    hipLaunchKernelGGL(copy_f32_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
  }
  return x2_check_launch("x2_copy_f32_multi");
}

// ---------------------------------------------------------------------------------- patches / tokens
// image fp32 (B,3,R,R) -> patch rows bf16 [B*g*g][3*ps*ps], column = c*ps*ps + py*ps + px (Conv2d weight order)
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ cols, int B, int R, int ps) {
  const int g = R / ps, kk = 3 * ps * ps;
  const long total = (long)B * g * g * kk / 4;
  long i = blockIdx.x * 256L + threadIdx.x;
  const long st = (long)gridDim.x * 256;
  for (; i < total; i += st) {
    const long e = i * 4;
    const int col = (int)(e % kk);
    const long prow = e / kk;
    const int px = col % ps, py = (col / ps) % ps, c = col / (ps * ps);
    const int pxg = (int)(prow % g), pyg = (int)((prow / g) % g), b = (int)(prow / ((long)g * g));
    const float4 v = *reinterpret_cast<const float4*>(img + (((long)b * 3 + c) * R + pyg * ps + py) * R + pxg * ps + px);
    *reinterpret_cast<u32x2*>(cols + e) = u32x2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
  }
}
extern "C" int x2_patchify(const float* image, void* cols, int B, int R, int ps, void* stream) {
  X2_REQUIRE(B > 0 && R > 0 && ps > 0 && R % ps == 0 && ps % 4 == 0, "x2_patchify: B=%d R=%d ps=%d", B, R, ps);
  const long total = (long)B * (R / ps) * (R / ps) * 3 * ps * ps / 4;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, image, (bf16_t*)cols, B, R, ps);
  return x2_check_launch("x2_patchify");
}

// x[b][0][:] = cls ; x[b][1+p][:] = patch[b*P+p][:]            (fp32, D % 4 == 0)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                              float* __restrict__ x, int B, int P, int D) {
  const long total = (long)B * (P + 1) * D / 4;
  long i = blockIdx.x * 256L + threadIdx.x;
  const long st = (long)gridDim.x * 256;
  for (; i < total; i += st) {
    const long e = i * 4;
    const int d = (int)(e % D);
    const long row = e / D;
    const int tkn = (int)(row % (P + 1)), b = (int)(row / (P + 1));
    const float4 v = tkn == 0 ? *reinterpret_cast<const float4*>(cls + d)
                              : *reinterpret_cast<const float4*>(patch + ((long)b * P + tkn - 1) * D + d);
    *reinterpret_cast<float4*>(x + e) = v;
  }
}
extern "C" int x2_assemble_tokens(const float* patch, const float* cls, float* x, int B, int P, int D, void* stream) {
  X2_REQUIRE(B > 0 && P > 0 && D % 4 == 0, "x2_assemble_tokens: B=%d P=%d D=%d", B, P, D);
  const long total = (long)B * (P + 1) * D / 4;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(assemble_tokens_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, patch, cls, x, B, P, D);
  return x2_check_launch("x2_assemble_tokens");
}

// backward of assemble: dpatch_bf16[b*P+p][:] = dx[b][1+p][:] ; dcls[:] += sum_b dx[b][0][:] (cls_grad_kernel below)
__global__ __launch_bounds__(256) void assemble_tokens_bwd_kernel(const float* __restrict__ dx, bf16_t* __restrict__ dpatch,
                                                                  float* dcls, int B, int P, int D) {
  const long total = (long)B * (P + 1) * D / 4;
  long i = blockIdx.x * 256L + threadIdx.x;
  const long st = (long)gridDim.x * 256;
  for (; i < total; i += st) {
    const long e = i * 4;
    const int d = (int)(e % D);
    const long row = e / D;
    const int tkn = (int)(row % (P + 1)), b = (int)(row / (P + 1));
    const float4 v = *reinterpret_cast<const float4*>(dx + e);
    if (tkn != 0) *reinterpret_cast<u32x2*>(dpatch + ((long)b * P + tkn - 1) * D + d) = u32x2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
  }
}
// dcls[:] += sum_b dx[b][0][:], the B rows added in ascending b by the one thread that owns the 4 columns (round 6: was one fp32 atomic per
// element from the kernel above - the only order-dependent sum of the vision tower's backward)
__global__ __launch_bounds__(64) void cls_grad_kernel(const float* __restrict__ dx, float* dcls, int B, long bstride, int D) {
  const int d = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (d >= D) return;
  float4 t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int b = 0; b < B; ++b) {
    const float4 v = *reinterpret_cast<const float4*>(dx + b * bstride + d);
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  float4 o = *reinterpret_cast<float4*>(dcls + d);
  o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
  *reinterpret_cast<float4*>(dcls + d) = o;
}
extern "C" int x2_assemble_tokens_bwd(const float* dx, void* dpatch, float* dcls, int B, int P, int D, void* stream) {
  X2_REQUIRE(B > 0 && P > 0 && D % 4 == 0, "x2_assemble_tokens_bwd: B=%d P=%d D=%d", B, P, D);
  const long total = (long)B * (P + 1) * D / 4;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(assemble_tokens_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dx, (bf16_t*)dpatch, dcls, B, P, D);
  hipLaunchKernelGGL(cls_grad_kernel, dim3((D / 4 + 63) / 64), dim3(64), 0, (hipStream_t)stream, dx, dcls, B, (long)(P + 1) * D, D);
  return x2_check_launch("x2_assemble_tokens_bwd");
}

// token 0 of every sample <- (weighted) mean of its patch tokens.  w: [B][P] weights or null (plain mean).
// fwd: x[b][0][:] = sum_p w[b][p] x[b][1+p][:] / sum_p w[b][p]
// bwd (add=1): g[b][1+p][:] += w[b][p] / sum_p w * g[b][0][:]  (token-0 row of g is then zeroed)
// Workgroup = 64 float4 columns x 4 waves; wave v walks patches p = v, v+4, ... (4 independent row streams per column
// slice instead of one serial walk over all P rows: 65 -> ~15 us on [64, 197, 768]); the forward combines the four
// partial sums through LDS.
__global__ __launch_bounds__(256) void pool_tokens_kernel(float* x, const float* __restrict__ w, int B, int P, int D, int bwd) {
  __shared__ float4 part[3][64];
  __shared__ float wpart[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int d = (blockIdx.y * 64 + lane) * 4;
  const bool live = d < D;
  float* xb = x + (long)b * (P + 1) * D;
  float wsum = (float)P;
  if (w) {                                   // every wave needs the total weight: lanes stride the row, waves share via LDS
    float s = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) s += w[(long)b * P + p];
    s = wave_sum(s);
    if (lane == 0) wpart[wave] = s;
    __syncthreads();
    wsum = wpart[0] + wpart[1] + wpart[2] + wpart[3];
  }
  const float inv = 1.f / wsum;
  if (!bwd) {
    float4 acc{0.f, 0.f, 0.f, 0.f};
    if (live)
      for (int p = wave; p < P; p += 4) {
        const float wp = w ? w[(long)b * P + p] : 1.f;
        const float4 v = *reinterpret_cast<const float4*>(xb + (long)(1 + p) * D + d);
        acc.x += wp * v.x; acc.y += wp * v.y; acc.z += wp * v.z; acc.w += wp * v.w;
      }
    if (wave > 0) part[wave - 1][lane] = acc;
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { const float4 o = part[k][lane]; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
      *reinterpret_cast<float4*>(xb + d) = float4{acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv};
    }
  } else {
    if (live) {
      const float4 g0 = *reinterpret_cast<const float4*>(xb + d);
      for (int p = wave; p < P; p += 4) {
        const float wp = (w ? w[(long)b * P + p] : 1.f) * inv;
        float4 v = *reinterpret_cast<float4*>(xb + (long)(1 + p) * D + d);
        v.x += wp * g0.x; v.y += wp * g0.y; v.z += wp * g0.z; v.w += wp * g0.w;
        *reinterpret_cast<float4*>(xb + (long)(1 + p) * D + d) = v;
      }
    }
    __syncthreads();                         // every wave has read the token-0 gradient before it is cleared
    if (wave == 0 && live) *reinterpret_cast<float4*>(xb + d) = float4{0.f, 0.f, 0.f, 0.f};
  }
}
extern "C" int x2_pool_tokens(float* x, const float* w, int B, int P, int D, int bwd, void* stream) {
  X2_REQUIRE(B > 0 && P > 0 && D % 4 == 0, "x2_pool_tokens: B=%d P=%d D=%d", B, P, D);
  hipLaunchKernelGGL(pool_tokens_kernel, dim3(B, (D + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, w, B, P, D, bwd);
  return x2_check_launch("x2_pool_tokens");
}

// ---------------------------------------------------------------------------------- rel-pos bias
// bias[h][i][j] = table[index[i][j]][h] for i,j < N (padded [H][N][ld]); biasT[h][j][i] likewise [H][N][ldT].
// indexT (optional) = the transposed index [j][i], built once by the host from the static buffer: with it the transposed
// copy is written row-contiguously as well (a thread-per-(i,j) scatter into biasT is one 4-byte write per cache line:
// 57 us per X2VLM-large block).
__global__ __launch_bounds__(256) void relpos_bias_kernel(const float* __restrict__ table, const long* __restrict__ index,
                                                          const long* __restrict__ indexT, float* bias, float* biasT, int N, int H,
                                                          int ld, int ldT, float scale) {
  const long total = (long)N * N;
  long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= total) return;
  const int i = (int)(e / N), j = (int)(e % N);
  const long idx = index[e];
  if (indexT && biasT) {
    const long idt = indexT[e];
    for (int h = 0; h < H; ++h) {
      bias[((long)h * N + i) * ld + j] = table[idx * H + h] * scale;
      biasT[((long)h * N + i) * ldT + j] = table[idt * H + h] * scale;
    }
    return;
  }
  for (int h = 0; h < H; ++h) {
    const float v = table[idx * H + h] * scale;
    bias[((long)h * N + i) * ld + j] = v;
    if (biasT) biasT[((long)h * N + j) * ldT + i] = v;
  }
}
extern "C" int x2_relpos_bias(const float* table, const long* index, const long* indexT, float* bias, float* biasT, int N, int H, int ld,
                              int ldT, float scale, void* stream) {
  X2_REQUIRE(N > 0 && H > 0 && ld >= N && (!biasT || ldT >= N), "x2_relpos_bias: N=%d H=%d ld=%d ldT=%d", N, H, ld, ldT);
  hipLaunchKernelGGL(relpos_bias_kernel, dim3((int)(((long)N * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, index, indexT,
                     bias, biasT, N, H, ld, ldT, scale);
  return x2_check_launch("x2_relpos_bias");
}
// dtable[index[i][j]][h] += sum_b dS[b][h][i][j]   (dS bf16 [B][H][N][ld], ld % 8 == 0), without atomics:
//   stage 1: the batch is cut into S slices; one thread sums 8 consecutive j of one (h, i) down its slice with 16-byte
//            loads and writes ws[s][h][i][j] (fp32);
//   stage 2: one wave per (table entry t, head h) gathers the positions that index t (CSR inverse of the static
//            relative_position_index: inv_off [T+1], inv_pos = i * ld + j) from all S slices and adds the total.
// (An atomic scatter - 465 k adds onto 8.8 k addresses per layer - took 3x the time of reading dS.)
__global__ __launch_bounds__(256) void relpos_bias_bsum_kernel(const bf16_t* __restrict__ dS, float* __restrict__ ws, int B, int N, int H,
                                                               int ld, int per) {
  const int per_row = ld >> 3;
  const long t = blockIdx.x * 256L + threadIdx.x;
  if (t >= (long)H * N * per_row) return;
  const int j0 = (int)(t % per_row) * 8;
  if (j0 >= N) return;
  const long hi = t / per_row;                       // h * N + i
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long bstride = (long)H * N * ld;
  const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
  const bf16_t* p = dS + hi * ld + j0;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(p + b * bstride);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += bf_lo(v[e]); acc[2 * e + 1] += bf_hi(v[e]); }
  }
  float* o = ws + ((long)blockIdx.y * H * N + hi) * ld + j0;
  *reinterpret_cast<float4*>(o) = float4{acc[0], acc[1], acc[2], acc[3]};
  *reinterpret_cast<float4*>(o + 4) = float4{acc[4], acc[5], acc[6], acc[7]};
}
__global__ __launch_bounds__(256) void relpos_bias_gather_kernel(const float* __restrict__ ws, const int* __restrict__ inv_off,
                                                                 const int* __restrict__ inv_pos, float* dtable, int T, int H, long plane,
                                                                 int S) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (w >= T * H) return;
  const int t = w / H, h = w % H;
  const int q0 = inv_off[t], q1 = inv_off[t + 1];
  float acc = 0.f;
  for (int q = q0 + lane; q < q1; q += 64) {
    const float* src = ws + (long)h * plane + inv_pos[q];
    for (int s_ = 0; s_ < S; ++s_) acc += src[(long)s_ * H * plane];
  }
  acc = wave_sum(acc);
  if (lane == 0) dtable[(long)t * H + h] += acc;
}
extern "C" int x2_relpos_bias_bwd(const void* dS, const int* inv_off, const int* inv_pos, float* dtable, int B, int N, int H, int ld,
                                  int T, float* ws, int slices, void* stream) {
  X2_REQUIRE(B > 0 && N > 0 && H > 0 && T > 0 && ld >= N && ld % 8 == 0 && slices >= 1 && slices <= B && ws && inv_off && inv_pos,
             "x2_relpos_bias_bwd: B=%d N=%d H=%d ld=%d T=%d slices=%d", B, N, H, ld, T, slices);
  const long threads = (long)H * N * (ld >> 3);
  const int per = (B + slices - 1) / slices;
  hipLaunchKernelGGL(relpos_bias_bsum_kernel, dim3((int)((threads + 255) / 256), slices), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dS, ws, B, N, H, ld, per);
  hipLaunchKernelGGL(relpos_bias_gather_kernel, dim3((T * H + 3) / 4), dim3(256), 0, (hipStream_t)stream, ws, inv_off, inv_pos, dtable, T,
                     H, (long)N * ld, slices);
  return x2_check_launch("x2_relpos_bias_bwd");
}

