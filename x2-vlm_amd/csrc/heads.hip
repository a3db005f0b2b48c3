// Embedding, gather/scatter, small fp32 linear, L2-normalise, softmax cross-entropy and
// hard-negative sampling kernels of the X^2-VLM step (gfx950).  These are the latency/HBM-bound
// ends of the path: BertEmbeddings (xbert.py:189-216), masked-position gather (xbert.py:1588-1589),
// projection heads + F.normalize (xvlm.py:785-792), ITC / ITM / MLM cross-entropies
// (xvlm.py:794-826, 895-899, xbert.py:1660-1661), torch.multinomial hard negatives (xvlm.py:828-857).
#include "x2_common.h"
#define SCATTER_MAX_R_E 8192        // rows marked per pass of the embedding backward's bitmap

// ------------------------------------------------------------------------------------ embeddings
// out[r][:] = word[ids[r]][:] + pos[r % L][:] + type0[:]      (fp32; the LayerNorm that follows is separate)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type0,
                                                        float* __restrict__ out, int R, int L, int D) {
  const int r = blockIdx.x;
  const long id = ids[r];
  const int l = r % L;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    const float4 a = *reinterpret_cast<const float4*>(word + id * D + d), b = *reinterpret_cast<const float4*>(pos + (long)l * D + d),
                 c = *reinterpret_cast<const float4*>(type0 + d);
    *reinterpret_cast<float4*>(out + (long)r * D + d) = float4{a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z, a.w + b.w + c.w};
  }
}
extern "C" int x2_embed_fwd(const long* ids, const float* word, const float* pos, const float* type0, float* out, int R, int L, int D,
                            void* stream) {
  X2_REQUIRE(R > 0 && L > 0 && D % 4 == 0, "x2_embed_fwd: R=%d L=%d D=%d", R, L, D);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, ids, word, pos, type0, out, R, L, D);
  return x2_check_launch("x2_embed_fwd");
}
// dword[ids[r]] += g[r] ; dpos[r % L] += g[r] ; dtype0 += g[r]   (rows ordered (sequence, position)).  No atomics (round 6: every sum in a fixed order,
// so two runs of a step give the same bits):
//   word rows: the workgroup of row r marks every row that carries the same token id in an LDS bitmap; if r is the FIRST of them it adds the marked rows in
//     ascending r and is the only writer of dword[ids[r]] ([CLS], [SEP], [MASK] and frequent words repeat hundreds of times per batch: the atomic form's result
//     depended on the order those adds arrived in);
//   position rows: one thread per (position, 4 columns) walks the batch; the per-position totals go to a small scratch row and
//   type-0 row: one thread per 4 columns adds the L totals in ascending position.
// (round 6, second form: the first one added a leader's rows one after the other - [MASK] is on ~600 rows of a base batch, ~1200 of a region
// batch: a chain of that many dependent row loads, 220 / 900 us.  Now the marked rows become a LIST in LDS, four row lanes of 256 threads walk it
// interleaved, eight rows in flight each; lane partials and the eight accumulators are combined in a fixed order: still no atomics, same bits
// on every run, 32 rows in flight instead of one.)
__global__ __launch_bounds__(1024) void embed_bwd_word_kernel(const long* __restrict__ ids, const float* __restrict__ g, float* dword, int R,
                                                              int D) {
  __shared__ uint32_t bm[SCATTER_MAX_R_E / 32];
  __shared__ uint16_t lst[SCATTER_MAX_R_E];
  __shared__ int woff[SCATTER_MAX_R_E / 32 + 1];
  __shared__ float4 part[3][256];
  const int r = blockIdx.x, tid = threadIdx.x, rl = tid >> 8, t = tid & 255;
  const long id = ids[r];
  for (int c0 = 0; c0 < D / 4; c0 += 256) {            // 1024 columns per round (D = 768 / 1024: one round)
    const int col = c0 + t;
    float4 tot{0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < R; r0 += SCATTER_MAX_R_E) {
      const int rn = min(SCATTER_MAX_R_E, R - r0), nw = (rn + 31) >> 5;
      __syncthreads();
      for (int w = tid; w < nw; w += 1024) bm[w] = 0u;
      __syncthreads();
      for (int q = tid; q < rn; q += 1024)
        if (ids[r0 + q] == id) atomicOr(&bm[q >> 5], 1u << (q & 31));
      __syncthreads();
      // exclusive prefix of the word populations (nw <= 256): Hillis-Steele over woff[1 .. nw]
      if (tid < 256) woff[tid + 1] = tid < nw ? __popc(bm[tid]) : 0;
      if (tid == 0) woff[0] = 0;
      __syncthreads();
      for (int st = 1; st < 256; st <<= 1) {
        int v = 0;
        if (tid < 256 && tid >= st) v = woff[tid + 1 - st];
        __syncthreads();
        if (tid < 256 && tid >= st) woff[tid + 1] += v;
        __syncthreads();
      }
      if (tid < nw) {
        uint32_t bits = bm[tid];
        int o = woff[tid];
        while (bits) { lst[o++] = (uint16_t)((tid << 5) + __builtin_ctz(bits)); bits &= bits - 1; }
      }
      __syncthreads();
      const int n = woff[256];
      if (n == 0) continue;
      if (r0 + (int)lst[0] < r) return;                  // an earlier row carries this id: that row's workgroup does the sum (block-uniform)
      if (col < D / 4) {
        float4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = float4{0.f, 0.f, 0.f, 0.f};
        for (int i = rl; i < n; i += 32) {               // row lane rl: list entries rl, rl + 4, ...; eight of them per trip
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int e = i + 4 * j;
            if (e < n) {
              const float4 v = *reinterpret_cast<const float4*>(g + (long)(r0 + lst[e]) * D + col * 4);
              acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
            }
          }
        }
        float4 p;
        p.x = ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
        p.y = ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
        p.z = ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
        p.w = ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
        if (rl > 0) part[rl - 1][t] = p;
        else tot.x += p.x, tot.y += p.y, tot.z += p.z, tot.w += p.w;
      }
      __syncthreads();
      if (rl == 0 && col < D / 4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float4 v = part[k][t]; tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w; }
      }
    }
    if (rl == 0 && col < D / 4) {
      float4* w = reinterpret_cast<float4*>(dword + id * D + col * 4);
      float4 o = *w;
      o.x += tot.x; o.y += tot.y; o.z += tot.z; o.w += tot.w;
      *w = o;
    }
  }
}
__global__ __launch_bounds__(256) void embed_bwd_pos_kernel(const float* __restrict__ g, float* dpos, float* tot, int R, int L, int D) {
  const int l = blockIdx.x, d = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (d >= D) return;
  float4 t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int r = l; r < R; r += L) {
    const float4 v = *reinterpret_cast<const float4*>(g + (long)r * D + d);
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  float* p = dpos + (long)l * D + d;
  p[0] += t.x; p[1] += t.y; p[2] += t.z; p[3] += t.w;          // one thread owns (l, d..d+3)
  *reinterpret_cast<float4*>(tot + (long)l * D + d) = t;
}
__global__ __launch_bounds__(64) void embed_bwd_type_kernel(const float* __restrict__ tot, float* dtype0, int L, int D) {
  const int d = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (d >= D) return;
  float4 t{0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) { const float4 v = *reinterpret_cast<const float4*>(tot + (long)l * D + d); t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
  float4 o = *reinterpret_cast<float4*>(dtype0 + d);
  o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
  *reinterpret_cast<float4*>(dtype0 + d) = o;
}
extern "C" int x2_embed_bwd(const long* ids, const float* g, float* dword, float* dpos, float* dtype0, int R, int L, int D, float* scratch,
                            void* stream) {
  X2_REQUIRE(R > 0 && L > 0 && D % 4 == 0 && D <= 4096 && scratch, "x2_embed_bwd: R=%d L=%d D=%d (D <= 4096, scratch of min(L, R) * D floats)", R, L, D);
  hipLaunchKernelGGL(embed_bwd_word_kernel, dim3(R), dim3(1024), 0, (hipStream_t)stream, ids, g, dword, R, D);
  const int Lp = L < R ? L : R;
  hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3(Lp, (D / 4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, dpos, scratch, R, L, D);
  hipLaunchKernelGGL(embed_bwd_type_kernel, dim3((D / 4 + 63) / 64), dim3(64), 0, (hipStream_t)stream, scratch, dtype0, Lp, D);
  return x2_check_launch("x2_embed_bwd");
}

// ------------------------------------------------------------------------------------ row gather / scatter-add
// dst[r][:] = src[idx[r]][:]  (rows of `len` floats, len % 4 == 0); optional bf16 copy.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* dst,
                                                          bf16_t* dstb, long len) {
  const long r = blockIdx.y;
  const float* s = src + (long)idx[r] * len;
  for (long e = (blockIdx.x * 256L + threadIdx.x) * 4; e < len; e += (long)gridDim.x * 1024) {
    const float4 v = *reinterpret_cast<const float4*>(s + e);
    if (dst) *reinterpret_cast<float4*>(dst + r * len + e) = v;
    if (dstb) *reinterpret_cast<u32x2*>(dstb + r * len + e) = u32x2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
  }
}
extern "C" int x2_gather_rows(const float* src, const int* idx, float* dst, void* dst_bf16, int R, long len, void* stream) {
  X2_REQUIRE(R > 0 && len > 0 && len % 4 == 0, "x2_gather_rows: R=%d len=%ld", R, len);
  const int bx = (int)((len / 4 + 255) / 256 < 64 ? (len / 4 + 255) / 256 : 64);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(bx, R), dim3(256), 0, (hipStream_t)stream, src, idx, dst, (bf16_t*)dst_bf16, len);
  return x2_check_launch("x2_gather_rows");
}
// Backward of the row gather: dst[d][:] = sum over {r : idx[r] == d} of src[r][:], EVERY dst row written (zeros where nothing points).
// One workgroup per (dst row, 1024-column chunk) marks the rows that point at d in an LDS bitmap (R <= 8192) and adds them in
// ascending r: no atomics (the first form - a zero fill of dst plus one fp32 atomic per element, 5.9 M of them for the 256
// sequences of the fusion pass - took 44 us per call in the tail segment, where nothing runs beside it), no dependence on the
// order workgroups finish in.
#define SCATTER_MAX_R 8192          // source rows marked per pass of the bitmap (any R: passes of 8192 rows, still ascending r)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst,
                                                           int R, long len, int chunks) {
  __shared__ uint32_t bm[SCATTER_MAX_R / 32];
  // (dst row, column chunk) folded into grid.x: no 65535-row limit of grid.y (the MLM gather backward has D = B * L rows)
  const int d = blockIdx.x / chunks;
  const long e = ((long)(blockIdx.x % chunks) * 256L + threadIdx.x) * 4;
  float4 acc{0.f, 0.f, 0.f, 0.f};
  for (int r0 = 0; r0 < R; r0 += SCATTER_MAX_R) {
    const int rn = min(SCATTER_MAX_R, R - r0), nw = (rn + 31) >> 5;
    __syncthreads();                                  // the previous pass's bitmap has been read by everyone
    for (int w = threadIdx.x; w < nw; w += 256) bm[w] = 0u;
    __syncthreads();
    for (int r = threadIdx.x; r < rn; r += 256)
      if (idx[r0 + r] == d) atomicOr(&bm[r >> 5], 1u << (r & 31));
    __syncthreads();
    if (e < len) {
      for (int w = 0; w < nw; ++w) {
        uint32_t bits = bm[w];
        while (bits) {
          const int r = r0 + (w << 5) + __builtin_ctz(bits);
          bits &= bits - 1;
          const float4 v = *reinterpret_cast<const float4*>(src + (long)r * len + e);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
    }
  }
  if (e < len) *reinterpret_cast<float4*>(dst + (long)d * len + e) = acc;
}
extern "C" int x2_scatter_rows(const float* src, const int* idx, float* dst, int R, int D, long len, void* stream) {
  X2_REQUIRE(R > 0 && D > 0 && len > 0 && len % 4 == 0, "x2_scatter_rows: R=%d D=%d len=%ld", R, D, len);
  const long chunks = (len / 4 + 255) / 256;
  X2_REQUIRE(chunks * D <= 2147483647L, "x2_scatter_rows: D=%d x %ld column chunks exceed the grid", D, chunks);
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)(chunks * D)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, R, len, (int)chunks);
  return x2_check_launch("x2_scatter_rows");
}

// ------------------------------------------------------------------------------------ small fp32 linear
// C[m][n] (+)= alpha * sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] (+ bias[n]); alpha read from device if alpha_ptr.
// 64x64 tile, 256 threads x (4x4) outputs, K step 16 through LDS.  For the heads (M <= a few hundred): their
// output grids are 1-12 workgroups, so the contraction is cut into gridDim.z slices whose partial tiles go to
// ws[z][m][n] and are added in slice order by linear_f32_reduce_kernel (deterministic: the ITC logits feed a loss
// that tests compare bit for bit between runs).
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* C,
                                                         const float* __restrict__ bias, const float* alpha_ptr, float alpha, int M, int N,
                                                         int K, long sam, long sak, long sbn, long sbk, long ldc, int accumulate, float* ws) {
  __shared__ float As[16][65], Bs[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kper = ((K + gridDim.z - 1) / gridDim.z + 15) & ~15;
  const int kbeg = blockIdx.z * kper, kend = min(K, kbeg + kper);
  float acc[4][4] = {};
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    for (int e = threadIdx.x; e < 1024; e += 256) {
      const int kk = e & 15, rr = e >> 4;
      const int m = m0 + rr, n = n0 + rr, k = k0 + kk;
      As[kk][rr] = (m < M && k < kend) ? A[m * sam + k * sak] : 0.f;
      Bs[kk][rr] = (n < N && k < kend) ? B[n * sbn + k * sbk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
  const float al = alpha_ptr ? alpha * alpha_ptr[0] : alpha;
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < M && n < N) {
        if (split) { ws[((long)blockIdx.z * M + m) * N + n] = al * acc[i][j]; continue; }
        float v = al * acc[i][j] + (bias ? bias[n] : 0.f);
        if (accumulate) v += C[m * ldc + n];
        C[m * ldc + n] = v;
      }
    }
}
__global__ __launch_bounds__(256) void linear_f32_reduce_kernel(const float* __restrict__ ws, float* C, const float* __restrict__ bias, int M,
                                                                int N, long ldc, int accumulate, int slices) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= (long)M * N) return;
  const int m = (int)(e / N), n = (int)(e % N);
  float v = bias ? bias[n] : 0.f;
  for (int z = 0; z < slices; ++z) v += ws[(long)z * M * N + e];
  if (accumulate) v += C[m * ldc + n];
  C[m * ldc + n] = v;
}
// ksplit > 1 needs ws of ksplit * M * N floats
extern "C" int x2_linear_f32(const float* A, const float* B, float* C, const float* bias, const float* alpha_ptr, float alpha, int M,
                             int N, int K, long sam, long sak, long sbn, long sbk, long ldc, int accumulate, int ksplit, float* ws,
                             void* stream) {
  X2_REQUIRE(M > 0 && N > 0 && K > 0 && ksplit >= 1 && ksplit <= 64, "x2_linear_f32: M=%d N=%d K=%d ksplit=%d", M, N, K, ksplit);
  X2_REQUIRE(ksplit == 1 || ws, "x2_linear_f32: ksplit=%d needs a workspace", ksplit);
  hipLaunchKernelGGL(linear_f32_kernel, dim3((N + 63) / 64, (M + 63) / 64, ksplit), dim3(256), 0, (hipStream_t)stream, A, B, C, bias,
                     alpha_ptr, alpha, M, N, K, sam, sak, sbn, sbk, ldc, accumulate, ws);
  if (ksplit > 1)
    hipLaunchKernelGGL(linear_f32_reduce_kernel, dim3((int)(((long)M * N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, C, bias, M, N,
                       ldc, accumulate, ksplit);
  return x2_check_launch("x2_linear_f32");
}

// ------------------------------------------------------------------------------------ L2 normalise rows
// y = x / max(||x||, 1e-12) ; bwd: dx = (dy - y * <dy, y>) / max(||x||, eps).  One wave per row.
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* out, int R, int D,
                                                     int bwd) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  float ss = 0.f, dot = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = x[(long)row * D + d]; ss += v * v; if (bwd) dot += v * dy[(long)row * D + d]; }
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), 1e-12f), inv = 1.f / nrm;
  if (!bwd) { for (int d = lane; d < D; d += 64) out[(long)row * D + d] = x[(long)row * D + d] * inv; return; }
  dot = wave_sum(dot) * inv * inv;   // <dy, y> / ||x||
  for (int d = lane; d < D; d += 64) out[(long)row * D + d] = dy[(long)row * D + d] * inv - x[(long)row * D + d] * dot * inv;
}
extern "C" int x2_l2norm(const float* x, const float* dy, float* out, int R, int D, int bwd, void* stream) {
  X2_REQUIRE(R > 0 && D > 0 && (!bwd || dy), "x2_l2norm: R=%d D=%d", R, D);
  hipLaunchKernelGGL(l2norm_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, out, R, D, bwd);
  return x2_check_launch("x2_l2norm");
}

// ------------------------------------------------------------------------------------ softmax cross-entropy
// One workgroup per row.  fwd: lse[r], loss_row[r] = lse - logit[label] (0 and not counted when label == ignore).
__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ labels, int C,
                                                     float* lse, float* loss_row) {
  __shared__ float sh[8];
  const int r = blockIdx.x;
  const float* z = logits + (long)r * ld;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, z[c]);
  mx = block_reduce(mx, sh, true);
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += __expf(z[c] - mx);
  s = block_reduce(s, sh, false);
  if (threadIdx.x == 0) {
    const float l = mx + logf(s);
    const long lab = labels[r];
    lse[r] = l;
    loss_row[r] = lab >= 0 ? l - z[lab] : 0.f;
  }
}
// loss = sum(loss_row) / count(label >= 0); out[0] = loss, out[1] = count
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ loss_row, const long* __restrict__ labels, int R, float* out) {
  __shared__ float sh[8];
  float s = 0.f, n = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) { s += loss_row[r]; n += labels[r] >= 0 ? 1.f : 0.f; }
  s = block_reduce(s, sh, false);
  n = block_reduce(n, sh, false);
  if (threadIdx.x == 0) { out[0] = s / n; out[1] = n; }
}
// dlogits[r][c] = (exp(z - lse) - [c == label]) * gscale * g[0] / count ; rows with ignored label -> 0 ; pad columns C..ldd-1 -> 0
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ g, const float* __restrict__ stat,
                                                     float gscale, int C, float* dlf, bf16_t* dlb, long ldd) {
  const int r = blockIdx.x;
  const long lab = labels[r];
  const float sc = lab >= 0 ? gscale * g[0] / stat[1] : 0.f;
  const float l = lse[r];
  const float* z = logits + (long)r * ld;
  for (int c = threadIdx.x; c < ldd; c += 256) {
    float v = 0.f;
    if (c < C && lab >= 0) v = (__expf(z[c] - l) - (c == lab ? 1.f : 0.f)) * sc;
    if (dlf) dlf[(long)r * ldd + c] = v;
    if (dlb) dlb[(long)r * ldd + c] = f2bf(v);
  }
}
// Fused MLM cross-entropy, second stage (first stage: x2_mlm_ce_fwd, gemm.hip): fold the (max, sum exp) pairs of a row's
// 64-column chunks into its log-partition; loss_row = lse - logit[label] (0 for ignored rows).  One wave per row.
__global__ __launch_bounds__(256) void ce_combine_kernel(const float* __restrict__ part, int chunks, const float* __restrict__ zlab,
                                                         const long* __restrict__ labels, int R, float* lse, float* loss_row) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float2* pr = reinterpret_cast<const float2*>(part) + (size_t)r * chunks;
  float mx = -INFINITY;
  for (int c = lane; c < chunks; c += 64) mx = fmaxf(mx, pr[c].x);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float s = 0.f;
  for (int c = lane; c < chunks; c += 64) { const float2 q = pr[c]; s += q.y * __expf(q.x - mx); }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    const float l = mx + logf(s);
    const long lab = labels[r];
    lse[r] = l;
    loss_row[r] = lab >= 0 ? l - zlab[r] : 0.f;
  }
}
extern "C" int x2_ce_combine(const float* part, int chunks, const float* zlab, const long* labels, int R, float* lse, float* loss_row,
                             float* out2, void* stream) {
  X2_REQUIRE(part && zlab && labels && lse && loss_row && out2 && R > 0 && chunks > 0, "x2_ce_combine: R=%d chunks=%d", R, chunks);
  hipLaunchKernelGGL(ce_combine_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, chunks, zlab, labels, R, lse, loss_row);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_row, labels, R, out2);
  return x2_check_launch("x2_ce_combine");
}
extern "C" int x2_ce_fwd(const float* logits, long ld, const long* labels, int R, int C, float* lse, float* loss_row, float* out2,
                         void* stream) {
  X2_REQUIRE(R > 0 && C > 0 && ld >= C, "x2_ce_fwd: R=%d C=%d ld=%ld", R, C, ld);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, C, lse, loss_row);
  hipLaunchKernelGGL(ce_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_row, labels, R, out2);
  return x2_check_launch("x2_ce_fwd");
}
extern "C" int x2_ce_bwd(const float* logits, long ld, const long* labels, const float* lse, const float* g, const float* stat,
                         float gscale, int R, int C, float* dl_f32, void* dl_bf16, long ldd, void* stream) {
  X2_REQUIRE(R > 0 && C > 0 && ldd >= C && (dl_f32 || dl_bf16), "x2_ce_bwd: R=%d C=%d ldd=%ld", R, C, ldd);
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, lse, g, stat, gscale, C, dl_f32,
                     (bf16_t*)dl_bf16, ldd);
  return x2_check_launch("x2_ce_bwd");
}

// ------------------------------------------------------------------------------------ hard negatives
// Row b: w[j] = softmax_j(sim[b][j]) + 1e-5, w[b] = 0 (or w[j] = 0 where group[j] == group[b]); draw one j
// with the uniform u[b] in [0,1) by inverse CDF.  One workgroup (<= 1024 columns) per row; no host sync.
__global__ __launch_bounds__(256) void sample_negatives_kernel(const float* __restrict__ sim, int n, const long* __restrict__ group,
                                                               const float* __restrict__ u, int* out) {
  __shared__ float sh[8];
  __shared__ float w[1024];
  const int b = blockIdx.x;
  const float* z = sim + (long)b * n;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < n; c += 256) mx = fmaxf(mx, z[c]);
  mx = block_reduce(mx, sh, true);
  float s = 0.f;
  for (int c = threadIdx.x; c < n; c += 256) s += __expf(z[c] - mx);
  s = block_reduce(s, sh, false);
  for (int c = threadIdx.x; c < n; c += 256) {
    const bool same = group ? group[c] == group[b] : c == b;
    w[c] = same ? 0.f : __expf(z[c] - mx) / s + 1e-5f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int c = 0; c < n; ++c) tot += w[c];
    const float t = u[b] * tot;
    float cum = 0.f; int pick = -1;
    for (int c = 0; c < n; ++c) { cum += w[c]; if (w[c] > 0.f && cum > t) { pick = c; break; } }
    if (pick < 0) for (int c = n - 1; c >= 0; --c) if (w[c] > 0.f) { pick = c; break; }
    // every candidate masked (all rows in one group): torch.multinomial raises in the reference (xvlm.py:845-855); the
    // host rejects that batch (XVLMBase.get_hard_negatives).  Never hand an out-of-range row index downstream.
    out[b] = pick < 0 ? b : pick;
  }
}
extern "C" int x2_sample_negatives(const float* sim, int n, const long* group, const float* u, int* out, void* stream) {
  X2_REQUIRE(n > 1 && n <= 1024, "x2_sample_negatives: n=%d not in (1,1024]", n);
  hipLaunchKernelGGL(sample_negatives_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, sim, n, group, u, out);
  return x2_check_launch("x2_sample_negatives");
}

// ------------------------------------------------------------------------------------ attention-mask / K-V sharing tables
// out[s][l] = (1 - atts[s][l]) * neg for l < L, 0 in the pad columns L..Lp-1: the additive key mask of BertModel
// (get_extended_attention_mask: neg = -10000; invert_attention_mask: -1e9; xbert.py:1105-1160) in the padded fp32 layout the
// attention kernels read.  One launch instead of five elementwise torch kernels per mask.
__global__ __launch_bounds__(256) void additive_mask_kernel(const long* __restrict__ atts, float* __restrict__ out, int S, int L, int Lp, float neg) {
  const long e = blockIdx.x * 256L + threadIdx.x;
  if (e >= (long)S * Lp) return;
  const int s_ = (int)(e / Lp), l = (int)(e % Lp);
  out[e] = l < L ? (1.0f - (float)atts[(long)s_ * L + l]) * neg : 0.f;
}
extern "C" int x2_additive_mask(const long* atts, float* out, int S, int L, int Lp, float neg, void* stream) {
  X2_REQUIRE(atts && out && S > 0 && L > 0 && Lp >= L, "x2_additive_mask: S=%d L=%d Lp=%d", S, L, Lp);
  hipLaunchKernelGGL(additive_mask_kernel, dim3((unsigned)(((long)S * Lp + 255) / 256)), dim3(256), 0, (hipStream_t)stream, atts, out, S, L, Lp, neg);
  return x2_check_launch("x2_additive_mask");
}
// CSR of "which query sequences use K/V batch b" from kv[s] (s < S, values in [0, Bi)): off[Bi + 1], order[S] = the sequence
// ids grouped by kv value, ascending inside a group (a stable counting sort: the same tables on every run).  Replaces
// argsort + scatter_add + cumsum (a dozen launches) - kv changes every step with the sampled hard negatives.  One workgroup.
__global__ __launch_bounds__(256) void kv_csr_kernel(const int* __restrict__ kv, int S, int Bi, int* __restrict__ off, int* __restrict__ order) {
  extern __shared__ int cnt[];                      // [Bi + 1]
  __shared__ int part[256];
  for (int b = threadIdx.x; b <= Bi; b += 256) cnt[b] = 0;
  __syncthreads();
  for (int e = threadIdx.x; e < S; e += 256) atomicAdd(&cnt[kv[e]], 1);
  __syncthreads();
  // exclusive scan of cnt[0..Bi): every thread sums a chunk, thread 0 scans the 256 chunk totals, chunks are rewritten
  const int per = (Bi + 255) / 256, b0 = threadIdx.x * per, b1 = min(Bi, b0 + per);
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += cnt[b];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int t = 0; t < 256; ++t) { const int v = part[t]; part[t] = run; run += v; } }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int b = b0; b < b1; ++b) { const int v = cnt[b]; cnt[b] = run; run += v; }
  __syncthreads();
  if (threadIdx.x == 0) cnt[Bi] = S;
  __syncthreads();
  for (int b = threadIdx.x; b <= Bi; b += 256) off[b] = cnt[b];
  // thread per K/V batch walks the sequences in increasing order: stable
  for (int b = threadIdx.x; b < Bi; b += 256) {
    int pos = cnt[b];
    const int end = cnt[b + 1];
    for (int e = 0; e < S && pos < end; ++e) if (kv[e] == b) order[pos++] = e;
  }
}
extern "C" int x2_kv_csr(const int* kv, int S, int Bi, int* off, int* order, void* stream) {
  X2_REQUIRE(kv && off && order && S > 0 && Bi > 0 && Bi <= 8192, "x2_kv_csr: S=%d Bi=%d (Bi <= 8192)", S, Bi);
  hipLaunchKernelGGL(kv_csr_kernel, dim3(1), dim3(256), (size_t)(Bi + 1) * sizeof(int), (hipStream_t)stream, kv, S, Bi, off, order);
  return x2_check_launch("x2_kv_csr");
}

// ------------------------------------------------------------------------------------ elementwise
// y = gelu(x) (fp32, exact erf) and its backward dx = dy * gelu'(x); used by the two small MLP heads
__global__ __launch_bounds__(256) void gelu_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* out, long n) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i < n) out[i] = dy ? dy[i] * dgelu_f(x[i]) : gelu_f(x[i]);
}
extern "C" int x2_gelu_f32(const float* x, const float* dy, float* out, long n, void* stream) {
  X2_REQUIRE(n > 0, "x2_gelu_f32: n=%ld", n);
  hipLaunchKernelGGL(gelu_f32_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, out, n);
  return x2_check_launch("x2_gelu_f32");
}
// out[n] += sum_m x[m][n]  (fp32 column sums: small-head bias gradients)
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, float* out, int M, int N) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += x[(long)m * N + n];
  out[n] += s;
}
extern "C" int x2_colsum_f32(const float* x, float* out, int M, int N, void* stream) {
  X2_REQUIRE(M > 0 && N > 0, "x2_colsum_f32: M=%d N=%d", M, N);
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, M, N);
  return x2_check_launch("x2_colsum_f32");
}

// ------------------------------------------------------------------------------------ glue of the step that used to be ATen launches
// Row tables of the pre-training step's 4B-row fusion batch (model_pretrain.py tail_losses; the reference runs these rows as four
// passes, xvlm.py:859-899 + model_pretrain.py:44-62): row r = q * B + b,
//   q = 0: (text b, image b)   q = 1: (text b, image ineg[b])   q = 2: (text tneg[b], image b)   q = 3: (masked text b = row B + b, image b)
// (with_match == 0: only the masked rows).  Writes t_idx[r] (row of the 2B-row text batch), kv[r] (image whose K / V row r attends
// to), the text attention mask of every row and its image attention mask.  One launch instead of ~12 (arange, add, 2 x cat,
// 2 x index, casts).
__global__ __launch_bounds__(256) void tail_index_kernel(const int* __restrict__ ineg, const int* __restrict__ tneg,
                                                         const long* __restrict__ text_atts, const long* __restrict__ image_atts, int B, int L,
                                                         int T, int with_match, int* __restrict__ t_idx, int* __restrict__ kv,
                                                         long* __restrict__ atts_out, long* __restrict__ enc_out) {
  const int r = blockIdx.x, q = with_match ? r / B : 3, b = r % B;
  const int ti = q == 2 ? tneg[b] : (q == 3 ? B + b : b), ki = q == 1 ? ineg[b] : b;
  if (threadIdx.x == 0) { t_idx[r] = ti; kv[r] = ki; }
  const long* ta = text_atts + (long)(ti % B) * L;
  for (int l = threadIdx.x; l < L; l += 256) atts_out[(long)r * L + l] = ta[l];
  const long* ia = image_atts + (long)ki * T;
  for (int t = threadIdx.x; t < T; t += 256) enc_out[(long)r * T + t] = ia[t];
}
extern "C" int x2_tail_index(const int* ineg, const int* tneg, const long* text_atts, const long* image_atts, int B, int L, int T,
                             int with_match, int* t_idx, int* kv, long* atts_out, long* enc_out, void* stream) {
  X2_REQUIRE(text_atts && image_atts && t_idx && kv && atts_out && enc_out && B > 0 && L > 0 && T > 0, "x2_tail_index: B=%d L=%d T=%d", B, L, T);
  X2_REQUIRE(!with_match || (ineg && tneg), "x2_tail_index: with_match needs the negative indices");
  hipLaunchKernelGGL(tail_index_kernel, dim3(with_match ? 4 * B : B), dim3(256), 0, (hipStream_t)stream, ineg, tneg, text_atts, image_atts, B, L, T,
                     with_match, t_idx, kv, atts_out, enc_out);
  return x2_check_launch("x2_tail_index");
}

// Stochastic depth (timm drop_path as the reference's Block uses it, beit2.py:205-207): one Bernoulli(1 - rate[l]) per block l,
// residual branch br and SAMPLE b; out[(l * 2 + br) * B * T + b * T + t] = keep / (1 - rate[l]) - the per-row factor the
// projection / fc2 GEMM epilogues multiply in.  Counter-based like the dropout masks (x2_common.h): the uniform of (l, br, b) is
// a hash of (seed [, device epoch word]), so a replayed hipGraph draws new keeps every step and the backward needs no mask tensor.
// Mirrored by kernels.droppath_keep() on the host.  One launch instead of rand + compare + cast + divide + repeat_interleave.
__global__ __launch_bounds__(256) void droppath_rows_kernel(const float* __restrict__ rates, uint32_t seed, const uint32_t* __restrict__ epoch,
                                                            int depth, int B, int T, float* __restrict__ out) {
  const long e = blockIdx.x * 256L + threadIdx.x, rows = (long)B * T;
  if (e >= 2L * depth * rows) return;
  const int lb = (int)(e / rows), b = (int)((e % rows) / T), l = lb >> 1;
  uint32_t s_ = seed;
  if (epoch) s_ = x2_hash(s_ + 0x9E3779B1u * epoch[0]);
  const uint32_t h = x2_hash(((uint32_t)lb * (uint32_t)B + (uint32_t)b) ^ s_);
  const float rate = rates[l], u = (float)(h >> 8) * (1.0f / 16777216.0f);
  out[e] = u >= rate ? 1.0f / (1.0f - rate) : 0.0f;
}
extern "C" int x2_droppath_rows(const float* rates, unsigned seed, const unsigned* epoch, int depth, int B, int T, float* out, void* stream) {
  X2_REQUIRE(rates && out && depth > 0 && B > 0 && T > 0, "x2_droppath_rows: depth=%d B=%d T=%d", depth, B, T);
  const long n = 2L * depth * B * T;
  hipLaunchKernelGGL(droppath_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rates, seed, epoch, depth, B, T, out);
  return x2_check_launch("x2_droppath_rows");
}

// Video path (xvlm.py:627-645, video_encoding 'avgpool'): out[b][t][:] = mean over the F frames of (x[b * F + f][t][:] + pos[f][:]).
// Backward: dx[b * F + f] = dy[b] / F for every frame, dpos[f][:] = sum over (b, t) of dy / F (the same row for every frame).
__global__ __launch_bounds__(256) void frame_mean_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos, float* __restrict__ out,
                                                             int Bc, int F, long TD, int D) {
  const long e = (blockIdx.x * 256L + threadIdx.x) * 4;
  if (e >= (long)Bc * TD) return;
  const long b = e / TD, o = e % TD;
  const int d = (int)(o % D);
  float4 a = float4{0.f, 0.f, 0.f, 0.f};
  for (int f = 0; f < F; ++f) {
    const float4 v = *reinterpret_cast<const float4*>(x + (b * F + f) * TD + o);
    float4 pp = float4{0.f, 0.f, 0.f, 0.f};
    if (pos) pp = *reinterpret_cast<const float4*>(pos + (long)f * D + d);
    a.x += v.x + pp.x; a.y += v.y + pp.y; a.z += v.z + pp.z; a.w += v.w + pp.w;
  }
  const float inv = 1.0f / (float)F;
  *reinterpret_cast<float4*>(out + e) = float4{a.x * inv, a.y * inv, a.z * inv, a.w * inv};
}
__global__ __launch_bounds__(256) void frame_mean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Bc, int F, long TD) {
  const long e = (blockIdx.x * 256L + threadIdx.x) * 4;
  if (e >= (long)Bc * TD) return;
  const long b = e / TD, o = e % TD;
  const float inv = 1.0f / (float)F;
  float4 g = *reinterpret_cast<const float4*>(dy + e);
  g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
  for (int f = 0; f < F; ++f) *reinterpret_cast<float4*>(dx + (b * F + f) * TD + o) = g;
}
// dpos[f][d] = (1 / F) sum over rows of dy[row][d]: one workgroup per 64 columns, 4 row slices, LDS fold
__global__ __launch_bounds__(256) void frame_pos_grad_kernel(const float* __restrict__ dy, float* __restrict__ dpos, long rows, int D, int F) {
  __shared__ float sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  float s = 0.f;
  if (c < D) for (long r = sl; r < rows; r += 4) s += dy[r * D + c];
  sh[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0 && c < D) {
    const float v = (sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]) / (float)F;
    for (int f = 0; f < F; ++f) dpos[(long)f * D + c] = v;
  }
}
extern "C" int x2_frame_mean(const float* x, const float* pos, const float* dy, float* out, float* dpos, int Bc, int F, int T, int D, int bwd,
                             void* stream) {
  X2_REQUIRE(Bc > 0 && F > 0 && T > 0 && D > 0 && D % 4 == 0, "x2_frame_mean: Bc=%d F=%d T=%d D=%d", Bc, F, T, D);
  const long TD = (long)T * D, n4 = (long)Bc * TD / 4;
  if (!bwd) {
    X2_REQUIRE(x && out, "x2_frame_mean: forward needs x and out");
    hipLaunchKernelGGL(frame_mean_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, pos, out, Bc, F, TD, D);
  } else {
    X2_REQUIRE(dy && out, "x2_frame_mean: backward needs dy and out (= dx)");
    hipLaunchKernelGGL(frame_mean_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, out, Bc, F, TD);
    if (dpos) hipLaunchKernelGGL(frame_pos_grad_kernel, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, dy, dpos, (long)Bc * T, D, F);
  }
  return x2_check_launch("x2_frame_mean");
}
