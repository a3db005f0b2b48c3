// bf16 MFMA GEMMs for gfx950 (MI355X): the dense contractions of the X^2-VLM step.
//
//   x2_gemm_nt : C[M,N] = epilogue(A[M,K] . B[N,K]^T)      forward linears and dgrads (on W^T copies)
//   x2_gemm_tn : C[N,K] (+)= A[Mc,N]^T . B[Mc,K]           weight gradients, grouped (one launch / layer)
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 each = 4x4 MFMA
// 16x16x32 bf16 tiles), contraction step 64, operands streamed HBM -> LDS with 16-byte
// global_load_lds (no VGPR round trip), double-buffered with one barrier per step, XOR-swizzled
// LDS images (swizzle applied on the per-lane SOURCE address, LDS image stays lane-linear),
// XCD-aware tile order.  The TN kernel reads its fragments with ds_read_b64_tr_b16, so neither
// activations nor gradients are ever transposed in HBM.
//
// Replaces (reference, all implicit ATen/cuBLAS calls): F.linear in beit2.py:131,160,62,66 and
// xbert.py:338-350,428,497,512,798,822 and their autograd backward.
#include "x2_common.h"
#include <type_traits>

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * 128)          // one operand tile: 128 rows x 64 bf16 (NT) or 64 rows x 128 bf16 (TN)
#define STAGE_BYTES (2 * TILE_BYTES)
#define GEMM_LDS_BYTES (2 * STAGE_BYTES)

struct GemmNT {
  const bf16_t* A; const bf16_t* B; void* C;
  const float* bias; const float* gamma; const float* resid; bf16_t* aux;
  int M, N, K;
  int lda, ldb, ldc, ldr, ldaux;
  int act;      // 0 none | 1 GELU (aux <- pre-activation) | 2 multiply by GELU'(aux)
  int out_f32;  // C is float (1) or bf16 (0)
  int group_m;  // tile raster: GROUP_M row-panels are walked column by column (L2 working set = GROUP_M A panels + a few B tiles)
  DropSpec drop;            // hidden dropout on (acc + bias) before the residual add (xbert.py:429, 513)
  const uint32_t* drop_epoch;  // device step counter mixed into drop.seed, or NULL (x2_common.h drop_at_epoch)
  const float* rowscale;    // [M] per-row factor before the residual add: DropPath keep/(1-p) per sample (beit2.py:206-207)
  float* colsum;            // [N] += column sums of the stored C (bias gradient of the layer below, fused)
  int dbg;                  // ablation switches, -DX2_PROBE builds only (probes/build_probe.sh): 4 = no epilogue, 16 = sc1 output stores;
                            // the shipped library compiles both out (NT_DBG below) and x2_tune(2, v != 0) is refused
  int ksplit;               // split-contraction launches (epilogue variant 7): columns of A / B per blockIdx.y slice
  // fused MLM cross-entropy (epilogue variants 8 / 9, x2_mlm_ce_fwd / _bwd): the logits never leave the accumulators
  const long* ce_labels;    // [M] target column or < 0 (ignored row)
  const float* ce_lse;      // [M] log-partition of every row (variant 9)
  const float* ce_g;        // [1] incoming loss gradient; ce_stat[1] = number of counted rows (variant 9)
  const float* ce_stat;
  float* ce_part;           // [M][N / 64][2] (max, sum exp(z - max)) of every 64-column chunk (variant 8)
  float* ce_zlab;           // [M] logit at the label (variant 8)
  float ce_gscale;
  int ce_C;                 // valid columns (vocabulary size; N is padded to a multiple of 64)
};

// logical tile id -> (row tile, col tile): XCD-contiguous chunks, inside a chunk groups of `gm` row panels
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
  const int gsz = gm * tiles_n, g = t / gsz, first = g * gm;
  const int rows = min(gm, tiles_m - first), r = t - g * gsz;
  tm = first + r % rows;
  tn = r / rows;
}

// ---- epilogue shared by the NT kernels ------------------------------------------------------------------
// The accumulators hold C^T fragments (lane = one row m, 4 consecutive n per 16x16 tile), which would store
// as 16 scattered 32-byte pieces per instruction.  Each wave instead parks its 64x64 fp32 sub-tile in LDS
// (two 32-row halves, rows padded to 68 floats: conflict-free both ways) and re-reads it row-major, so that
// 16 lanes cover one 256-byte row segment: full-line loads of the residual / saved pre-activation and
// full-line stores of C and aux.  bias / GELU / GELU' / dropout / layer-scale / DropPath / residual are applied
// on the way out; optional column sums (bias gradient of the layer below) leave as one atomic per column.
// Caller guarantees (barrier) that no wave still reads operand tiles from `smem`.
// 16-byte global store; wt = write-through + do not keep the line in this XCD's L2 (sc1): outputs are consumed by
// a later kernel, keeping them resident only evicts the operand panels the other workgroups of the XCD re-read.
__device__ __forceinline__ void st16(void* p, u32x4 v, bool wt) {
  if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else *reinterpret_cast<u32x4*>(p) = v;
}

// VAR: the epilogue's feature set as a compile-time constant for the shapes that carry the step (a runtime flag tested inside
// the row-group loop costs ~1 % of the NT GEMM time each: two extra ablation flags measured +2.3 % on the whole step):
//   0 bias -> bf16            (qkv / q / kv projections, bf16 input gradients)
//   1 bias -> fp32            (fp32 input gradients, logits, head linears)
//   2 bias, GELU, pre-activation saved -> bf16                  (fc1 / intermediate forward)
//   3 x GELU'(saved pre-activation) -> bf16                     (input gradient through the GELU; its bias-gradient column
//     sums are a separate two-stage kernel: fused as one atomic per column and wave they cost 30-50 us per launch)
//   5 bias, (dropout), + residual -> fp32                       (BERT output projections, BERT input gradients)
//   6 bias, layer scale (x DropPath row factor), + residual -> fp32   (BEiT proj / fc2; the value before the scale is NOT saved:
//     the layer-scale backward needs no activation, rowwise.hip x2_layerscale_finish)
//   4 everything decided at run time (any other combination)
//   7 partial products of one contraction slice -> fp32 workspace (x2_gemm_nt_splitk; slice = blockIdx.y)
//   8 bias, then per row and 64-column chunk (max, sum exp) + the logit at the label: softmax statistics, nothing stored
//   9 bias, then (softmax - onehot) * row scale -> bf16: gradient of the mean cross-entropy w.r.t. the logits
//  10 = 3 plus the column sums of the result (bias gradient of fc1 / intermediate): every wave adds up its rows and writes ONE
//     partial row colsum[2 * row tile + wave row][N]; the caller reduces the partial rows (x2_reduce_partials*): replaces a
//     stand-alone pass over the [M, 4D] gradient (x2_colsum_bf16: 77 MB read per vision block) without the atomics that
//     made the first fused form slower than that pass
#ifdef X2_PROBE
#define NT_DBG(p, bit) (((p).dbg & (bit)) != 0)
// per-phase time stamps of gemm_nt256_kernel (probe builds only): wave 0 of every workgroup writes wall_clock64() (100 MHz) at
// kernel entry, after the prologue's first barrier (operands of step 0 landed), after the last contraction step and after the
// epilogue to g_nt_probe[blockIdx.x][4]; probes/nt_phase_times.py sets the buffer through x2_probe_set_buffer
static __device__ unsigned long long* g_nt_probe = nullptr;
#define NT_STAMP(i) do { if (g_nt_probe && threadIdx.x == 0) g_nt_probe[(size_t)blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
// loop ablations of gemm_nt256_kernel (x2_tune(2, bits)): 32 = the loop requests no operand tiles (the prologue's are re-read), 64 = no fragment
// reads after step 0 (stale registers), 128 = no MFMAs - which PAIR of the loop's three activities costs the time (probes/nt_phase_times.py)
#define NT_ABL_DECL(p) const bool abl_dma_ = NT_DBG(p, 32), abl_rd_ = NT_DBG(p, 64), abl_mm_ = NT_DBG(p, 128); bool rd_ = true
#define NT_ABL_NO_DMA(kt) if (abl_dma_ && (kt) >= 2) return
#define NT_ABL_NO_READ() if (!rd_) return
#define NT_ABL_NO_MFMA() if (abl_mm_) return
#define NT_ABL_STEP(kt) if (abl_rd_ && (kt) > 0) rd_ = false
#else
#define NT_ABL_DECL(p)
#define NT_ABL_NO_DMA(kt)
#define NT_ABL_NO_READ()
#define NT_ABL_NO_MFMA()
#define NT_ABL_STEP(kt)
#define NT_DBG(p, bit) false
#define NT_STAMP(i) do { } while (0)
#endif
// One 32-row half of a wave tile into its staging rows (row-major, 68-float pitch).  Two accumulator shapes:
//   f32x4 acc[TM][4]   - 16x16x32 MFMA tiles (C^T: a lane holds row lane & 15 and 4 consecutive columns at (lane >> 4) * 4 of each 16-column tile)
//   f32x16 acc[TM/2][2] - 32x32x16 MFMA tiles (C^T: a lane holds row lane & 31 and, of each 32-column tile, the columns 8 g + 4 (lane >> 5) + 0..3, g < 4)
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int TM>
__device__ __forceinline__ void epi_stage_half(f32x4 (&acc)[TM][4], int half, bool full, float* stg, int lane) {
  const int frow = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (full || ii == 0) *reinterpret_cast<f32x4*>(stg + (ii * 16 + frow) * 68 + j * 16 + fg * 4) = acc[full ? half * 2 + ii : half * 2][j];
}
template <int RT>
__device__ __forceinline__ void epi_stage_half(f32x16 (&acc)[RT][2], int half, bool, float* stg, int lane) {
  const int row = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(stg + row * 68 + tj * 32 + g * 8 + h * 4) =
          f32x4{acc[half][tj][4 * g], acc[half][tj][4 * g + 1], acc[half][tj][4 * g + 2], acc[half][tj][4 * g + 3]};
}
template <int V> struct EpiTraits {
  static constexpr bool generic = V == 4;
  static constexpr int act = V == 2 ? 1 : (V == 3 || V == 10) ? 2 : 0;
  static constexpr bool out_f32 = V == 1 || V == 5 || V == 6 || V == 7;
  static constexpr bool resid = V == 5 || V == 6;
  static constexpr bool scale = V == 6;           // gamma and optional rowscale
  static constexpr bool aux0 = false;             // act == 0 with aux (value before the layer scale saved): generic feature set only
  static constexpr bool drop = V == 5;            // dropout possible (still a runtime test on thr16, outside the hot variants)
  static constexpr bool colsum = false;
  static constexpr bool colparts = V == 10;       // column sums of the stored values as one partial row per wave row (no atomics)
};
template <int TM, int VAR, class Acc>
__device__ __forceinline__ void nt_epilogue(const GemmNT& p, Acc& acc, char* smem, int wave, int lane, int mw0, int nw0, int prow = 0) {
  using E = EpiTraits<VAR>;
  const int act = E::generic ? p.act : E::act;
  const bool out_f32 = E::generic ? p.out_f32 != 0 : E::out_f32;
  const bool has_resid = E::generic ? p.resid != nullptr : E::resid;
  const bool has_scale = E::generic ? (p.gamma != nullptr || p.rowscale != nullptr) : E::scale;
  const bool has_aux0 = E::generic ? (p.act == 0 && p.aux != nullptr) : E::aux0;
  const bool has_drop = (E::generic || E::drop) ? p.drop.thr16 != 0 : false;
  const bool has_colsum = E::generic ? p.colsum != nullptr : (E::colsum && p.colsum != nullptr);
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 68);
  // read-back: 8 lanes x 8 columns cover one 64-column row (32-byte fp32 reads, 16-byte bf16 / 2 x 16-byte fp32 stores),
  // 8 rows per instruction: half the store instructions of a 4-column mapping (the tail is store-issue bound)
  const int er = lane >> 3, ec = (lane & 7) * 8;
  const int n = nw0 + ec;
  const bool nok = n < p.N;           // N % 8 == 0 is required by the host wrapper when this path is used
  float bb[8], gg[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bb[e] = 0.f; gg[e] = 1.f; }
  if (nok && p.bias) { const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
    bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w; }
  if (nok && has_scale && p.gamma) { const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + n), g1 = *reinterpret_cast<const float4*>(p.gamma + n + 4);
    gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w; }
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool wt = NT_DBG(p, 16);
  const DropSpec drop_ = drop_at_epoch(p.drop, p.drop_epoch);
#pragma unroll
  for (int half = 0; half < (TM + 1) / 2; ++half) {
    // an odd TM ends with a 16-row half: `full` folds at compile time once the loop is unrolled
    const bool full = half * 2 + 1 < TM;
    // operands the epilogue reads from HBM (residual rows, saved pre-activation) are requested for all four row groups
    // of this half BEFORE the accumulators go through LDS: their latency hides behind the staging round trip instead
    // of sitting in front of every row group's arithmetic (rows / columns past the edge are clamped, results discarded).
    // Same-box A/B on the whole step: 27.84 -> 27.30 ms (probes/run_ab_lib.sh).
    float4 r0[4], r1[4];
    u32x4 pre[4];
    const int nc = nok ? n : 0;
    if (has_resid) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (!full && rr >= 2) continue;
        const int mc = min(mw0 + half * 32 + rr * 8 + er, p.M - 1);
        r0[rr] = *reinterpret_cast<const float4*>(p.resid + (size_t)mc * p.ldr + nc);
        r1[rr] = *reinterpret_cast<const float4*>(p.resid + (size_t)mc * p.ldr + nc + 4);
      }
    }
    if (act == 2) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (!full && rr >= 2) continue;
        const int mc = min(mw0 + half * 32 + rr * 8 + er, p.M - 1);
        pre[rr] = *reinterpret_cast<const u32x4*>(p.aux + (size_t)mc * p.ldaux + nc);
      }
    }
    epi_stage_half(acc, half, full, stg, lane);
    // same-wave LDS traffic only: no barrier, the compiler's lgkmcnt wait orders write -> read
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (!full && rr >= 2) continue;
      const int row = rr * 8 + er;
      const int m = mw0 + half * 32 + row;
      const float4 a0 = *reinterpret_cast<const float4*>(stg + row * 68 + ec), a1 = *reinterpret_cast<const float4*>(stg + row * 68 + ec + 4);
      if (m >= p.M || !nok) continue;
      float v[8] = {a0.x + bb[0], a0.y + bb[1], a0.z + bb[2], a0.w + bb[3], a1.x + bb[4], a1.y + bb[5], a1.z + bb[6], a1.w + bb[7]};
      if constexpr (VAR == 8) {
        // the 8 lanes sharing `er` hold the 64 columns of row m: chunk statistics by three xor-shuffles inside that group
        const long lab = p.ce_labels[m];
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = n + e < p.ce_C ? fmaxf(mx, v[e]) : mx;
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        float se = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          se += n + e < p.ce_C ? __expf(v[e] - mx) : 0.f;
          if ((long)(n + e) == lab) p.ce_zlab[m] = v[e];
        }
        se += __shfl_xor(se, 1, 64); se += __shfl_xor(se, 2, 64); se += __shfl_xor(se, 4, 64);
        if ((lane & 7) == 0) *reinterpret_cast<float2*>(p.ce_part + ((size_t)m * (p.N >> 6) + (nw0 >> 6)) * 2) = float2{mx, se};
        continue;
      }
      if constexpr (VAR == 9) {
        const long lab = p.ce_labels[m];
        const float sc = lab >= 0 ? p.ce_gscale * p.ce_g[0] / p.ce_stat[1] : 0.f, l = p.ce_lse[m];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = n + e < p.ce_C ? (__expf(v[e] - l) - ((long)(n + e) == lab ? 1.f : 0.f)) * sc : 0.f;
        st16(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n,
             u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])}, false);
        continue;
      }
      if (act == 1) {
        st16(p.aux + (size_t)m * p.ldaux + n, u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])}, wt);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = gelu_f(v[r]);
      } else if (act == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[2 * r] *= dgelu_f(bf_lo(pre[rr][r])); v[2 * r + 1] *= dgelu_f(bf_hi(pre[rr][r])); }
      } else if (has_aux0) {
        st16(p.aux + (size_t)m * p.ldaux + n, u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])}, wt);
      }
      if (has_drop) {
        float dm[4];
        drop_mul4(drop_, (uint32_t)m * (uint32_t)p.N + (uint32_t)n, dm);
        v[0] *= dm[0]; v[1] *= dm[1]; v[2] *= dm[2]; v[3] *= dm[3];
        drop_mul4(drop_, (uint32_t)m * (uint32_t)p.N + (uint32_t)n + 4u, dm);
        v[4] *= dm[0]; v[5] *= dm[1]; v[6] *= dm[2]; v[7] *= dm[3];
      }
      if (has_scale) {
        const float rs_ = p.rowscale ? p.rowscale[m] : 1.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= gg[r] * rs_;
      }
      if (has_resid) {
        v[0] += r0[rr].x; v[1] += r0[rr].y; v[2] += r0[rr].z; v[3] += r0[rr].w;
        v[4] += r1[rr].x; v[5] += r1[rr].y; v[6] += r1[rr].z; v[7] += r1[rr].w;
      }
      if (has_colsum || E::colparts) {
#pragma unroll
        for (int r = 0; r < 8; ++r) cs[r] += v[r];
      }
      if (out_f32) {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
        st16(c, u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, wt);
        st16(c + 4, u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, wt);
      } else {
        st16(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n,
             u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])}, wt);
      }
    }
  }
  if constexpr (E::colparts) {     // this wave's rows, folded over the 8 lanes that share a column group: one 256-byte partial-row segment
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs[e] += __shfl_xor(cs[e], 8, 64); cs[e] += __shfl_xor(cs[e], 16, 64); cs[e] += __shfl_xor(cs[e], 32, 64); }
    if (er == 0 && nok) {
      float* dst = p.colsum + (size_t)prow * p.N + n;
      *reinterpret_cast<float4*>(dst) = float4{cs[0], cs[1], cs[2], cs[3]};
      *reinterpret_cast<float4*>(dst + 4) = float4{cs[4], cs[5], cs[6], cs[7]};
    }
  }
  if (has_colsum) {     // lanes sharing (lane & 7) hold the same 8 columns for different rows: fold 8 -> 1, one atomic per column
#pragma unroll
    for (int e = 0; e < 8; ++e) { cs[e] += __shfl_xor(cs[e], 8, 64); cs[e] += __shfl_xor(cs[e], 16, 64); cs[e] += __shfl_xor(cs[e], 32, 64); }
    if (er == 0 && nok) {
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(p.colsum + n + e, cs[e]);
    }
  }
}

// ---- staged epilogue for fp32 outputs with ROW-CONTIGUOUS stores (the only form of feature sets 1, 5, 6 since round 4) ----
// nt_epilogue gives a lane 8 consecutive columns of one row; for fp32 outputs that is two 16-byte stores per lane whose
// pieces interleave at a 32-byte stride: every store instruction half-fills 16 lines (8 rows x 2).  Here a lane takes 4
// columns of TWO rows (r and r + 4): a store instruction writes 4 rows x 256 contiguous bytes = 8 full lines; the residual
// is loaded the same way.  Feature sets 1, 5, 6 (the fp32-out ones), same arithmetic.
template <int TM, int VAR, class Acc>
__device__ __forceinline__ void nt_epilogue_f4(const GemmNT& p, Acc& acc, char* smem, int wave, int lane, int mw0, int nw0) {
  using E = EpiTraits<VAR>;
  static_assert(!E::generic && E::out_f32 && E::act == 0 && VAR <= 6, "nt_epilogue_f4: feature sets 1, 5, 6");
  const bool has_drop = E::drop ? p.drop.thr16 != 0 : false;
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 68);
  const int er = lane >> 4, ec = (lane & 15) * 4;          // rows er and er + 4 of every 8-row group, 4 columns
  const int n = nw0 + ec;
  const bool nok = n < p.N;                                // N % 8 == 0
  float bb[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {1.f, 1.f, 1.f, 1.f};
  if (nok && p.bias) { const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n); bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; }
  if (E::scale && nok && p.gamma) { const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + n); gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; }
  const DropSpec drop_ = drop_at_epoch(p.drop, p.drop_epoch);
#pragma unroll
  for (int half = 0; half < (TM + 1) / 2; ++half) {
    const bool full = half * 2 + 1 < TM;
    float4 rs[4][2];
    const int nc = nok ? n : 0;
    if (E::resid) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        if (!full && rr >= 2) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int mc = min(mw0 + half * 32 + rr * 8 + h * 4 + er, p.M - 1);
          rs[rr][h] = *reinterpret_cast<const float4*>(p.resid + (size_t)mc * p.ldr + nc);
        }
      }
    }
    epi_stage_half(acc, half, full, stg, lane);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (!full && rr >= 2) continue;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = rr * 8 + h * 4 + er;
        const int m = mw0 + half * 32 + row;
        const float4 a0 = *reinterpret_cast<const float4*>(stg + row * 68 + ec);
        if (m >= p.M || !nok) continue;
        float v[4] = {a0.x + bb[0], a0.y + bb[1], a0.z + bb[2], a0.w + bb[3]};
        if (has_drop) {
          float dm[4];
          drop_mul4(drop_, (uint32_t)m * (uint32_t)p.N + (uint32_t)n, dm);
          v[0] *= dm[0]; v[1] *= dm[1]; v[2] *= dm[2]; v[3] *= dm[3];
        }
        if (E::scale) {
          const float rs_ = p.rowscale ? p.rowscale[m] : 1.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= gg[r] * rs_;
        }
        if (E::resid) { v[0] += rs[rr][h].x; v[1] += rs[rr][h].y; v[2] += rs[rr][h].z; v[3] += rs[rr][h].w; }
        st16(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n,
             u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, false);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NT: both operands K-contiguous.  LDS image per operand: [128 rows][8 chunks of 16 B], chunk c of
// row r stored at chunk position c ^ (r & 7)  -> conflict-free ds_read_b128 fragment reads.
// ---------------------------------------------------------------------------------------------
// TM: 16-row MFMA tiles per wave along M: 4 -> 128x128 block tile, 6 -> 192x128 (2 x 80 KB LDS = exactly 2 blocks / CU);
// F4: fp32 outputs stored row-contiguously (nt_epilogue_f4; feature sets 1, 5, 6 only)
template <int TM, int VAR, bool F4 = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNT p) {
  constexpr int BMT = 32 * TM;                       // block rows (2 waves along M)
  if constexpr (VAR == 7) {                          // slice blockIdx.y of the contraction -> its own M x N partial in the workspace
    const int koff = blockIdx.y * p.ksplit;
    p.A += koff; p.B += koff;
    p.K = min(p.ksplit, p.K - koff);
    p.C = reinterpret_cast<float*>(p.C) + (size_t)blockIdx.y * p.M * p.ldc;
  }
  constexpr int A_BYTES = BMT * 128, STG = A_BYTES + TILE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BMT - 1) / BMT;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, p.group_m, tm, tn);
  const int m0 = tm * BMT, n0 = tn * BN;

  // per-thread global sources for the TM+4 chunks this thread stages per K-step
  const bf16_t* srcA[TM]; const bf16_t* srcB[4];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int q = i * 256 + tid, row = q >> 3, c = (q & 7) ^ (row & 7);
    int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
    srcA[i] = p.A + (size_t)ra * p.lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid, row = q >> 3, c = (q & 7) ^ (row & 7);
    int rb = n0 + row; rb = rb < p.N ? rb : p.N - 1;
    srcB[i] = p.B + (size_t)rb * p.ldb + c * 8;
  }
  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STG;
#pragma unroll
    for (int i = 0; i < TM; ++i) glds16(srcA[i] + (size_t)kt * BK, base + (i * 256 + wave * 64) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(srcB[i] + (size_t)kt * BK, base + A_BYTES + (i * 256 + wave * 64) * 16);
  };

  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = lds_addr(smem);
  const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
  const uint32_t offA = (uint32_t)((wm * 16 * TM + frow) * 128);
  const uint32_t offB = (uint32_t)(A_BYTES + (wn * 64 + frow) * 128);

  const int nk = p.K / BK;
  // Software pipeline (one barrier per K-step, placed in the MIDDLE of the step):
  //   fragments of k-half 1 are requested before the MFMAs of k-half 0 are issued, and the fragments of the NEXT
  //   tile's k-half 0 before the MFMAs of k-half 1, so every LDS read has 4*TM MFMAs (>= 256 cycles) to land behind;
  //   at the barrier each wave has all of tile kt in registers, so slot kt&1 is refilled (tile kt+2) right after it:
  //   a global->LDS request still has a full K-step to complete with only two slots.
  bf16x8 a0[TM], b0[4], a1[TM], b1[4];
  auto frags = [&](uint32_t sb, int ks, bf16x8 (&a)[TM], bf16x8 (&b)[4]) {
    const uint32_t cs = (uint32_t)(((ks * 4 + fg) ^ fsw) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = lds_read_b128(sb + offB + j * 2048 + cs);
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = lds_read_b128(sb + offA + i * 2048 + cs);
  };
  // MFMAs [first, last) of the 4*TM that consume one fragment set (operands swapped: the accumulator tile is C^T,
  // i.e. a lane holds m = frow and 4 consecutive n)
  auto mma = [&](bf16x8 (&a)[TM], bf16x8 (&b)[4], auto first, auto last) {
#pragma unroll
    for (int t = decltype(first)::value; t < decltype(last)::value; ++t)
      acc[t >> 2][t & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[t & 3], a[t >> 2], acc[t >> 2][t & 3], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using IN = std::integral_constant<int, 4 * TM>;
  // One K-step.  Each fragment set is requested right after the FIRST MFMA of the previous set, so the only LDS wait
  // (the compiler's lgkmcnt(0) in front of that first MFMA) finds nothing outstanding that was not issued 4*TM-1
  // MFMAs earlier.  The loop is peeled (refill / has-next are compile-time) to keep the body branch-free.
  auto step = [&](int kt, auto refill, auto has_next) {
    mma(a0, b0, I0{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    frags(lds0 + (kt & 1) * STG, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma(a0, b0, I1{}, IN{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(has_next)::value) {
      // tile kt is in registers (lgkmcnt) and tile kt+1 has landed (vmcnt) for this wave; after the barrier, for all
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if constexpr (decltype(refill)::value) stage(kt + 2, kt & 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1, I0{}, I1{});
      __builtin_amdgcn_sched_barrier(0);
      frags(lds0 + ((kt + 1) & 1) * STG, 0, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1, I1{}, IN{});
    } else {
      mma(a1, b1, I0{}, IN{});
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (nk > 1) stage(1, 1);
  frags(lds0, 0, a0, b0);
  int kt = 0;
  for (; kt + 2 < nk; ++kt) step(kt, std::true_type{}, std::true_type{});
  if (kt + 1 < nk) { step(kt, std::false_type{}, std::true_type{}); ++kt; }
  step(kt, std::false_type{}, std::false_type{});

  __syncthreads();                                   // every wave is done reading the last operand tiles
  if constexpr (F4) {
    nt_epilogue_f4<TM, VAR>(p, acc, smem, wave, lane, m0 + wm * 16 * TM, n0 + wn * 64);
    return;
  }
  if (NT_DBG(p, 4)) { if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.C)[0] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3]; return; }
  nt_epilogue<TM, VAR>(p, acc, smem, wave, lane, m0 + wm * 16 * TM, n0 + wn * 64, tm * 2 + wm);
}

// ---------------------------------------------------------------------------------------------
// NT, 256-column tiles on the schedule of the 256x256 weight-gradient kernel below (gemm_tn256_kernel): (32 * TMW) x 256
// outputs per 512-thread workgroup - 8 waves as 2 (rows) x 4 (columns), each a (16 * TMW) x 64 wave tile = TMW x 4 MFMA
// tiles (TMW = 8: 128 accumulator VGPRs) - one workgroup per CU.  Why: the 128x128 kernel above stages 1/64 byte per
// flop through L2 -> LDS and reads 0.5 LDS fragments per MFMA; its main loops run at 0.8-1.07 PFLOP/s, bound by that
// traffic and not by MFMA issue.  A 256x256 tile stages 1/128 byte per flop, a 128x64 wave tile reads 0.375 fragments
// per MFMA (the weight-gradient kernel's main loop: 1.24 PFLOP/s).
//   LDS: 2 contraction steps x 4 half-tiles (A0, A1 = the two wave rows' 16*TMW rows of A; B0, B1 = columns n0.., n0+128..)
//   x 16 KB, each a [rows][64 k] image, chunk c (16 B) of row r at position c ^ (r & 7) as in gemm_nt_kernel.
//   One contraction step = 4 phases (quadrants rows-lo x cols-lo, rows-lo x cols-hi, rows-hi x cols-hi, rows-hi x cols-lo
//   of the wave tile), every phase also requests ONE half-tile:
//     phase 1, 2: A0, A1 of step t+1 into the other buffer (last read in step t-1, behind that step's end barrier)
//     phase 3, 4: B0, B1 of step t+2 into THIS buffer (every wave holds its B fragments of step t after phase 2: mid barrier)
//   so 2-4 half-tiles are always in flight across the two barriers of a step: counted s_waitcnt vmcnt(4), never 0 in the loop.
// TMW = 8 / 7 / 6 / 5 (256 / 224 / 192 / 160 rows): the host picks the height whose tile count fills whole rounds of the
// 256 CUs best (x2_gemm_nt: nt256_plan) - at these sizes a launch is 1-3 rounds, so the last round's fill decides.
// Epilogue: nt_epilogue<TMW, VAR> as above (the wave's 64 columns x 16*TMW rows through its 8.5 KB of LDS staging).
// ---------------------------------------------------------------------------------------------
#define N2_HALF 16384
#define N2_LDS_BYTES (8 * N2_HALF)
template <int TMW, int VAR>
__global__ __launch_bounds__(512) void gemm_nt256_kernel(GemmNT p) {
  constexpr int BMT = 32 * TMW, HROWS = 16 * TMW;       // block rows, rows of one A half (= of one wave row)
  constexpr int TH = (TMW + 1) / 2;                     // row tiles of the first quadrant pair (rows-lo); rows-hi = TMW - TH
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + 255) / 256, tiles_m = (p.M + BMT - 1) / BMT;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, p.group_m, tm, tn);
  const int m0 = tm * BMT, n0 = tn * 256;

  // per-thread global sources: half h of A / B, chunk i (16 B each; a B half is 128 rows x 8 chunks = 2 x 512, an A half
  // 16*TMW rows x 8 = 128*TMW chunks: the second pass only covers waves below 2*TMW - 8)
  const bf16_t* srcA[2][2]; const bf16_t* srcB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = i * 512 + tid, row = q >> 3, c = (q & 7) ^ (row & 7);
      int ra = m0 + h * HROWS + (row < HROWS ? row : HROWS - 1); ra = ra < p.M ? ra : p.M - 1;
      int rb = n0 + h * 128 + row; rb = rb < p.N ? rb : p.N - 1;
      srcA[h][i] = p.A + (size_t)ra * p.lda + c * 8;
      srcB[h][i] = p.B + (size_t)rb * p.ldb + c * 8;
    }
  const bool a2 = wave < 2 * TMW - 8;                   // this wave takes part in the second pass over an A half
  NT_ABL_DECL(p);
  auto issue = [&](int kt, auto slot) {                 // slot 0, 1: A halves; 2, 3: B halves
    constexpr int S = decltype(slot)::value;
    NT_ABL_NO_DMA(kt);
    char* base = smem + ((kt & 1) * 4 + S) * N2_HALF + wave * 1024;
    if constexpr (S < 2) {
      glds16(srcA[S][0] + (size_t)kt * BK, base);
      if (TMW == 8 || a2) glds16(srcA[S][1] + (size_t)kt * BK, base + 8192);
    } else {
      glds16(srcB[S - 2][0] + (size_t)kt * BK, base);
      glds16(srcB[S - 2][1] + (size_t)kt * BK, base + 8192);
    }
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;

  f32x4 acc[TMW][4];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = lds_addr(smem);
  const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
  const uint32_t offA = (uint32_t)(wr * N2_HALF + frow * 128);
  const uint32_t offB = (uint32_t)((2 + (wc >> 1)) * N2_HALF + ((wc & 1) * 64 + frow) * 128);
  bf16x8 fa[2][TH], fb[2][4];                           // [k-half of the step][tile]: one row group of A, all of B
  auto readA = [&](uint32_t buf, auto hi_) {            // row tiles [0, TH) or [TH, TMW)
    constexpr bool hi = decltype(hi_)::value;
    NT_ABL_NO_READ();
    constexpr int first = hi ? TH : 0, count = hi ? TMW - TH : TH;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < count; ++i)
        fa[ks][i] = lds_read_b128(buf + offA + (first + i) * 2048 + (uint32_t)(((ks * 4 + fg) ^ fsw) << 4));
  };
  auto readB = [&](uint32_t buf, int jh) {              // column tiles 2*jh, 2*jh + 1
    NT_ABL_NO_READ();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[ks][jh * 2 + j] = lds_read_b128(buf + offB + (jh * 2 + j) * 2048 + (uint32_t)(((ks * 4 + fg) ^ fsw) << 4));
  };
  auto quad = [&](auto hi_, auto jh_) {                 // one quadrant: row group x column pair, both k-halves
    constexpr bool hi = decltype(hi_)::value;
    constexpr int jh = decltype(jh_)::value, first = hi ? TH : 0, count = hi ? TMW - TH : TH;
    NT_ABL_NO_MFMA();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < count; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          // operands swapped: the accumulator tile is C^T (a lane holds m = frow and 4 consecutive n), as nt_epilogue expects
          acc[first + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][jh * 2 + j], fa[ks][i], acc[first + i][jh * 2 + j], 0, 0, 0);
  };
  using LO = std::false_type; using HI = std::true_type;
#define N2_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto ktile = [&](int kt, auto n1_, auto n2_) {        // n1: step kt+1 exists, n2: step kt+2 exists
    constexpr bool n1 = decltype(n1_)::value, n2 = decltype(n2_)::value;
    const uint32_t buf = lds0 + (uint32_t)((kt & 1) * 4 * N2_HALF);
    NT_ABL_STEP(kt);
    // phase 1
    readA(buf, LO{}); readB(buf, 0); readB(buf, 1);
    if constexpr (n1) issue(kt + 1, S0{});
    N2_FENCE();
    quad(LO{}, S0{}); N2_FENCE();
    // phase 2
    if constexpr (n1) issue(kt + 1, S1{});
    N2_FENCE();
    quad(LO{}, S1{}); N2_FENCE();
    // phase 3: every wave holds its B fragments of this step before B0 / B1 of this buffer are refilled
    if constexpr (n2) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); issue(kt + 2, S2{}); }
    readA(buf, HI{});
    N2_FENCE();
    quad(HI{}, S1{}); N2_FENCE();
    // phase 4
    if constexpr (n2) issue(kt + 2, S3{});
    N2_FENCE();
    quad(HI{}, S0{}); N2_FENCE();
    if constexpr (n1) {
      // step kt+1 complete in LDS for this wave (only B0 / B1 of step kt+2 may still be in flight); then for all waves
      if constexpr (n2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      N2_FENCE();
    }
  };
  const int nk = p.K / BK;
  NT_STAMP(0);
  issue(0, S0{}); issue(0, S1{}); issue(0, S2{}); issue(0, S3{});
  if (nk > 1) { issue(1, S2{}); issue(1, S3{}); asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  NT_STAMP(1);
  int kt = 0;
  for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{}, std::true_type{});
  if (kt + 1 < nk) { ktile(kt, std::true_type{}, std::false_type{}); ++kt; }
  ktile(kt, std::false_type{}, std::false_type{});
#undef N2_FENCE
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave is done reading operand tiles
  NT_STAMP(2);
  if (NT_DBG(p, 4)) { if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.C)[0] = acc[1][1][1] + acc[2][2][2] + acc[3][3][3]; return; }
  nt_epilogue<TMW, VAR>(p, acc, smem, wave, lane, m0 + wr * HROWS, n0 + wc * 64);
#ifdef X2_PROBE
  __builtin_amdgcn_s_waitcnt(0);                                         // the wave's stores have left before the last stamp
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  NT_STAMP(3);
#endif
}

// ---------------------------------------------------------------------------------------------
// The same kernel at its most-used height (TMW = 5: 160 x 256 tiles) on a THREE-stage operand ring (round 5).  Why: the loop ablation
// (profiles/r09i_nt256_loop_ablation.txt) shows the operand DMA - not the matrix pipe, not the fragment reads - as the slowest single
// activity of a contraction step (0.84 us alone against 0.78 for the MFMAs and 1.18 for the step), at 40 % of the LDS-DMA path's rate: it is
// latency-bound at the 32-64 KB the two-stage ring keeps in flight (A halves one step ahead, B halves two).  At this height a stage is
// 2 x 10 KB (A halves: 80 rows) + 2 x 16 KB (B halves) = 52 KB, so three stages fit the CU's 160 KB: EVERY half-tile is requested two
// steps ahead (52-104 KB in flight), into the stage that was read in the previous step - behind that step's end barrier, so the
// mid-step barrier of the two-stage form is gone as well: one barrier per contraction step.
//   step t reads stage t % 3; its four phases request A0, A1, B0, B1 of step t + 2 into stage (t + 2) % 3; at the end of the step
//   s_waitcnt vmcnt(6) (a wave issues 6 requests per step, waves 0 and 1 eight: the count retires everything issued BEFORE this step,
//   i.e. step t + 1's operands) and the barrier make step t + 1 readable.
// ---------------------------------------------------------------------------------------------
#define N3_A_HALF 10240
#define N3_STAGE (2 * N3_A_HALF + 2 * N2_HALF)
#define N3_LDS_BYTES (3 * N3_STAGE)
// PIPE: the fragment reads of step t + 1's first two phases are issued in the LAST phase of step t, behind the (moved) barrier, so that no
// phase begins with a wait for its own LDS reads (all eight waves of the workgroup run in lockstep: without this the LDS read burst at the
// head of a step and the matrix pipe take turns).
template <int VAR, bool PIPE>
__global__ __launch_bounds__(512) void gemm_nt256s3_kernel(GemmNT p) {
  constexpr int TMW = 5, BMT = 32 * TMW, HROWS = 16 * TMW, TH = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + 255) / 256, tiles_m = (p.M + BMT - 1) / BMT;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, p.group_m, tm, tn);
  const int m0 = tm * BMT, n0 = tn * 256;
  const bf16_t* srcA[2][2]; const bf16_t* srcB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = i * 512 + tid, row = q >> 3, c = (q & 7) ^ (row & 7);
      int ra = m0 + h * HROWS + (row < HROWS ? row : HROWS - 1); ra = ra < p.M ? ra : p.M - 1;
      int rb = n0 + h * 128 + row; rb = rb < p.N ? rb : p.N - 1;
      srcA[h][i] = p.A + (size_t)ra * p.lda + c * 8;
      srcB[h][i] = p.B + (size_t)rb * p.ldb + c * 8;
    }
  const bool a2 = wave < 2 * TMW - 8;                   // rows 64..79 of an A half: the second pass of waves 0, 1
  auto issue = [&](int stage, int kt, auto slot) {      // slot 0, 1: A halves; 2, 3: B halves
    constexpr int S = decltype(slot)::value;
    char* base = smem + stage * N3_STAGE + (S < 2 ? S * N3_A_HALF : 2 * N3_A_HALF + (S - 2) * N2_HALF) + wave * 1024;
    if constexpr (S < 2) {
      glds16(srcA[S][0] + (size_t)kt * BK, base);
      if (a2) glds16(srcA[S][1] + (size_t)kt * BK, base + 8192);
    } else {
      glds16(srcB[S - 2][0] + (size_t)kt * BK, base);
      glds16(srcB[S - 2][1] + (size_t)kt * BK, base + 8192);
    }
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
  f32x4 acc[TMW][4];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t lds0 = lds_addr(smem);
  const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
  const uint32_t offA = (uint32_t)(wr * N3_A_HALF + frow * 128);
  const uint32_t offB = (uint32_t)(2 * N3_A_HALF + (wc >> 1) * N2_HALF + ((wc & 1) * 64 + frow) * 128);
  bf16x8 fa[2][TH], fb[2][4];
  bf16x8 fah[2][TMW - TH], fbx[2][2];                   // PIPE: rows-hi fragments of A in registers of their own; next step's B columns 0, 1
  auto readA = [&](uint32_t buf, auto hi_) {
    constexpr bool hi = decltype(hi_)::value;
    constexpr int first = hi ? TH : 0, count = hi ? TMW - TH : TH;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < count; ++i) {
        const bf16x8 v = lds_read_b128(buf + offA + (first + i) * 2048 + (uint32_t)(((ks * 4 + fg) ^ fsw) << 4));
        if constexpr (PIPE && hi) fah[ks][i] = v; else fa[ks][i] = v;
      }
  };
  auto readB = [&](uint32_t buf, int jh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[ks][jh * 2 + j] = lds_read_b128(buf + offB + (jh * 2 + j) * 2048 + (uint32_t)(((ks * 4 + fg) ^ fsw) << 4));
  };
  auto readBx = [&](uint32_t buf) {                     // PIPE: B columns 0, 1 of the NEXT step while this step's are still multiplied
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fbx[ks][j] = lds_read_b128(buf + offB + j * 2048 + (uint32_t)(((ks * 4 + fg) ^ fsw) << 4));
  };
  auto quad = [&](auto hi_, auto jh_) {
    constexpr bool hi = decltype(hi_)::value;
    constexpr int jh = decltype(jh_)::value, first = hi ? TH : 0, count = hi ? TMW - TH : TH;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < count; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[first + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][jh * 2 + j], (PIPE && hi) ? fah[ks][i] : fa[ks][i],
                                                                               acc[first + i][jh * 2 + j], 0, 0, 0);
  };
  using LO = std::false_type; using HI = std::true_type;
#define N3_FENCE() __builtin_amdgcn_sched_barrier(0)
  int rs = 0, ws = 2;                                   // stage read by this step, stage that receives step kt + 2
  // PIPE form of a step.  On entry the fragments of phases 1 and 2 (A rows-lo, B columns 0..3) are in registers (fa, fb; B columns
  // 0, 1 arrive in fbx and are moved over first).  Phase 2 also fetches A rows-hi; after phase 3 the wave waits for everything it
  // requested before this step's B0 (vmcnt(4): A0, A1, B0 of step kt+2 may be in flight) - i.e. step kt+1's operands - and the barrier makes
  // them readable; phase 4 fetches step kt+1's fragments behind its MFMAs.  One barrier per step, no phase starts with an LDS wait of its own.
  auto ptile = [&](int kt, auto n1_, auto n2_, auto first_) {
    constexpr bool n1 = decltype(n1_)::value, n2 = decltype(n2_)::value, first = decltype(first_)::value;
    const uint32_t buf = lds0 + (uint32_t)(rs * N3_STAGE), nbuf = lds0 + (uint32_t)((rs == 2 ? 0 : rs + 1) * N3_STAGE);
    if constexpr (!first) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      N3_FENCE();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { fb[ks][0] = fbx[ks][0]; fb[ks][1] = fbx[ks][1]; }
    }
    if constexpr (n2) issue(ws, kt + 2, S0{});
    N3_FENCE();
    quad(LO{}, S0{}); N3_FENCE();
    readA(buf, HI{});
    if constexpr (n2) issue(ws, kt + 2, S1{});
    N3_FENCE();
    quad(LO{}, S1{}); N3_FENCE();
    if constexpr (n2) issue(ws, kt + 2, S2{});
    N3_FENCE();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    N3_FENCE();
    quad(HI{}, S1{}); N3_FENCE();
    if constexpr (n1) {
      if constexpr (n2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      N3_FENCE();
      readA(nbuf, LO{}); readB(nbuf, 1); readBx(nbuf);
    }
    if constexpr (n2) issue(ws, kt + 2, S3{});
    N3_FENCE();
    quad(HI{}, S0{}); N3_FENCE();
    rs = rs == 2 ? 0 : rs + 1;
    ws = ws == 2 ? 0 : ws + 1;
  };
  auto ktile = [&](int kt, auto n1_, auto n2_) {        // n1: step kt+1 exists, n2: step kt+2 exists
    constexpr bool n1 = decltype(n1_)::value, n2 = decltype(n2_)::value;
    const uint32_t buf = lds0 + (uint32_t)(rs * N3_STAGE);
    readA(buf, LO{}); readB(buf, 0); readB(buf, 1);
    if constexpr (n2) issue(ws, kt + 2, S0{});
    N3_FENCE();
    quad(LO{}, S0{}); N3_FENCE();
    if constexpr (n2) issue(ws, kt + 2, S1{});
    N3_FENCE();
    quad(LO{}, S1{}); N3_FENCE();
    readA(buf, HI{});
    if constexpr (n2) issue(ws, kt + 2, S2{});
    N3_FENCE();
    quad(HI{}, S1{}); N3_FENCE();
    if constexpr (n2) issue(ws, kt + 2, S3{});
    N3_FENCE();
    quad(HI{}, S0{}); N3_FENCE();
    if constexpr (n1) {
      // everything this wave requested BEFORE this step (= step kt+1's operands) has landed; then for all waves.  lgkmcnt(0): this
      // wave's fragment reads of the stage are done before another wave may refill it (next step's requests go to stage rs + 2 = the
      // one read in THIS step only after the NEXT barrier, so this is belt and braces - the MFMAs above consumed every fragment)
      if constexpr (n2) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      N3_FENCE();
    }
    rs = rs == 2 ? 0 : rs + 1;
    ws = ws == 2 ? 0 : ws + 1;
  };
  const int nk = p.K / BK;
  issue(0, 0, S0{}); issue(0, 0, S1{}); issue(0, 0, S2{}); issue(0, 0, S3{});
  if (nk > 1) {
    issue(1, 1, S0{}); issue(1, 1, S1{}); issue(1, 1, S2{}); issue(1, 1, S3{});
    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  int kt = 0;
  if constexpr (PIPE) {
    readA(lds0, LO{}); readB(lds0, 0); readB(lds0, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    N3_FENCE();
    if (nk >= 3) { ptile(0, std::true_type{}, std::true_type{}, std::true_type{}); kt = 1; }
    else if (nk == 2) { ptile(0, std::true_type{}, std::false_type{}, std::true_type{}); kt = 1; }
    else { ptile(0, std::false_type{}, std::false_type{}, std::true_type{}); kt = 1; }
    for (; kt + 2 < nk; ++kt) ptile(kt, std::true_type{}, std::true_type{}, std::false_type{});
    if (kt + 1 < nk) { ptile(kt, std::true_type{}, std::false_type{}, std::false_type{}); ++kt; }
    if (kt < nk) ptile(kt, std::false_type{}, std::false_type{}, std::false_type{});
  } else {
    for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{}, std::true_type{});
    if (kt + 1 < nk) { ktile(kt, std::true_type{}, std::false_type{}); ++kt; }
    ktile(kt, std::false_type{}, std::false_type{});
  }
#undef N3_FENCE
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave is done reading operand tiles
  nt_epilogue<TMW, VAR>(p, acc, smem, wave, lane, m0 + wr * HROWS, n0 + wc * 64);
}

// ---------------------------------------------------------------------------------------------
// NT, 256-column tiles, "ping-pong" schedule on 32x32x16 MFMAs (round 6).  The kernels above run their eight waves in lockstep: every
// wave reads fragments, then every wave multiplies, so on each SIMD the matrix pipe and the LDS / DMA issue take turns (46 % MFMA issue
// inside the loop, profiles/r09b).  Here the two waves that share a SIMD (wave w and w + 4) work in opposite segments of a contraction
// step: while the waves of row 0 multiply, the waves of row 1 fetch their fragments and request operand tiles, and vice versa - one
// barrier between segments, the row-1 waves one barrier behind for the whole tile.  A segment holds ALL MFMAs of one contraction step
// of the wave (RT x 2 tiles of 32x32 x 4 k-slices of 16: 16-24 instructions of 32 cycles = 512-768 matrix-pipe cycles, issued under
// s_setprio 1 with nothing between them), so the pipe sees back-to-back MFMAs from alternating waves and everything else is issued in
// the partner's shadow.  The 32x32x16 form halves the MFMA count per flop (half the issue slots) at the same fragment bytes.
//   Tile: 32 (RT0 + RT1) rows x 256 columns; wave row 0 owns RT0 32-row tiles, wave row 1 RT1 (RT0 <= RT1: the 160-row tile is 2 + 3 -
//   the pipe is time-shared, so unequal halves cost nothing as long as each fetch segment fits in the partner's multiply segment;
//   row 0 runs first and requests the larger B tiles during row 1's longer multiply segment); wave column c owns columns 64 c .. + 63.
//   LDS: 2 stages x {A0 (RT0 x 4 KB), A1 (RT1 x 4 KB), B0, B1 (16 KB each)}, images [rows][64 k] with chunk c (16 B) of row r at
//   position c ^ ((r >> 1) & 7): conflict-free ds_read_b128 for the 32-row fragment (lanes = rows r .. r + 31 of ONE chunk).
//   DMA: buffer_load ... lds through buffer descriptors (rows past M / N read zeros: no clamping, one 32-bit offset per request).
//   Row 0 (phases 2t: fetch, 2t + 1: multiply) requests B(t + 1) at the head of its fetch segment and waits for it behind its multiply
//   segment; row 1 (phases 2t + 1, 2t + 2) requests A1(t + 1) and A0(t + 2) and waits for A0(t + 1) behind its fetch segment, for
//   A1(t + 1) behind its multiply segment: every request has >= one multiply segment (>= 512 cycles) + most of a fetch segment to land.
// Epilogue: nt_epilogue / nt_epilogue_f4 through the f32x16 staging overload.
// ---------------------------------------------------------------------------------------------
#define PP_BHALF 16384
template <int N_> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
template <int RT0, int RT1, int VAR, int NST>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(GemmNT p) {
  constexpr int BMT = 32 * (RT0 + RT1);
  constexpr int A0_BYTES = RT0 * 4096, A1_BYTES = RT1 * 4096, STAGE = A0_BYTES + A1_BYTES + 2 * PP_BHALF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + 255) / 256, tiles_m = (p.M + BMT - 1) / BMT;
  int tm, tn;
  tile_coords(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, p.group_m, tm, tn);
  const int m0 = tm * BMT, n0 = tn * 256;
  const uint32_t lds0 = lds_addr(smem);
  // this lane's chunk of a DMA request: LDS image is lane-linear, request q of a region covers rows 32 q + 8 wc + (lane >> 3)
  const int drow = wc * 8 + (lane >> 3), dchunk = (lane & 7) ^ ((wc & 1) * 4 + (lane >> 4));
  // fragment address pieces: row (lane & 31) of a 32-row tile, k-slice ks -> chunk (2 ks + (lane >> 5)) ^ swizzle(row)
  uint32_t foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = (uint32_t)((lane & 31) * 128 + (((ks * 2 + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
  const uint32_t boff = (uint32_t)(A0_BYTES + A1_BYTES + (wc >> 1) * PP_BHALF + (wc & 1) * 8192);
  const int nk = p.K / BK;
  NT_ABL_DECL(p);
  NT_STAMP(0);

  auto run = [&](auto rt_, auto first_) {
    constexpr int RT = decltype(rt_)::value;
    constexpr bool FIRST = decltype(first_)::value;
    const uint32_t aoff = FIRST ? 0u : (uint32_t)A0_BYTES;
    f32x16 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bf16x8 fa[RT][4], fb[2][4];
    // operand requests.  Row 0 owns B (two halves of 128 weight rows), row 1 owns A (both row groups)
    const __amdgpu_buffer_rsrc_t rs = FIRST
        ? __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)(((size_t)(p.N - 1) * p.ldb + p.K) * 2), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((size_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const int ld = FIRST ? p.ldb : p.lda;
    const uint32_t v0 = (uint32_t)(((size_t)((FIRST ? n0 : m0) + drow) * ld + dchunk * 8) * 2);     // row `drow` of the tile
    const uint32_t step32 = (uint32_t)ld * 64u;                                                     // 32 rows further, bytes
    auto dma = [&](int kt, int stage, int region_bytes_off, int row32_first, auto count_) {        // `count` requests of 32 rows each
      constexpr int CNT = decltype(count_)::value;
      NT_ABL_NO_DMA(kt + 1);
#pragma unroll
      for (int q = 0; q < CNT; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + stage * STAGE + region_bytes_off + q * 4096 + wc * 1024),
                                                 16, v0 + (uint32_t)(row32_first + q) * step32, kt * 128, 0, 0);
    };
    using C4 = std::integral_constant<int, 4>; using CR0 = std::integral_constant<int, RT0>; using CR1 = std::integral_constant<int, RT1>;
    auto reads = [&](uint32_t sb) {
      NT_ABL_NO_READ();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j][ks] = lds_read_b128(sb + boff + foff[ks] + j * 4096);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < RT; ++i) fa[i][ks] = lds_read_b128(sb + aoff + foff[ks] + i * 4096);
    };
    auto mfmas = [&]() {
      NT_ABL_NO_MFMA();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            // operands swapped: the accumulator tile is C^T (a lane holds m = lane & 31 and 4 x 4 consecutive n)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
#define PP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PP_BARRIER() do { PP_FENCE(); asm volatile("s_barrier" ::: "memory"); PP_FENCE(); } while (0)
    if constexpr (NST == 3) {
      // Three whole stages (tile heights whose 3 stages fit 160 KB): every region of step t + 2 is requested at the head of step t's fetch segment, into
      // the stage read in step t - 1.  Row 0 waits for B(t + 1) behind its multiply segment of step t (requested two steps = four segments
      // earlier), row 1 for A(t + 1) behind its fetch segment of step t (three segments).  Measured and NOT kept (profiles/r12a_nt_pp_per_shape.txt (3),
      // r12b_nt_pp_prefetch.txt; code at commit f9489e1): the requests spread between the fragment reads or between the MFMAs instead of one burst
      // (+-1 %), and an L2 prefetch of the A tile's lines six steps ahead through the same texture path (+10 %: it costs as many line requests as it saves).
      constexpr int MINE = FIRST ? 8 : RT0 + RT1;        // requests of this wave per step
      auto request = [&](int kt, int stage) {
        if constexpr (FIRST) { dma(kt, stage, A0_BYTES + A1_BYTES, 0, C4{}); dma(kt, stage, A0_BYTES + A1_BYTES + PP_BHALF, 4, C4{}); }
        else { dma(kt, stage, 0, 0, CR0{}); dma(kt, stage, A0_BYTES, RT0, CR1{}); }
      };
      request(0, 0);
      if (nk > 1) { request(1, 1); pp_wait_vm<MINE>(); } else pp_wait_vm<0>();
      PP_BARRIER();
      if constexpr (!FIRST) PP_BARRIER();               // row 1 runs one segment behind
      NT_STAMP(1);
      int rs = 0, ws = 2;
      for (int t = 0; t < nk; ++t) {
        const uint32_t sb = lds0 + (uint32_t)(rs * STAGE);
        NT_ABL_STEP(t);
        const bool more = t + 2 < nk;
        if (more) request(t + 2, ws);
        reads(sb);
        PP_FENCE();
        if constexpr (!FIRST) { if (more) pp_wait_vm<MINE>(); else pp_wait_vm<0>(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        PP_BARRIER();
        mfmas();
        PP_FENCE();
        if constexpr (FIRST) { if (more) pp_wait_vm<MINE>(); else pp_wait_vm<0>(); }
        PP_BARRIER();
        rs = rs == 2 ? 0 : rs + 1;
        ws = ws == 2 ? 0 : ws + 1;
      }
    } else {
      // prologue: step 0 complete in LDS (row 1 also has A0 of step 1 under way)
      if constexpr (FIRST) {
        dma(0, 0, A0_BYTES + A1_BYTES, 0, C4{}); dma(0, 0, A0_BYTES + A1_BYTES + PP_BHALF, 4, C4{});
        pp_wait_vm<0>();
      } else {
        dma(0, 0, A0_BYTES, RT0, CR1{}); dma(0, 0, 0, 0, CR0{});
        if (nk > 1) { dma(1, 1, 0, 0, CR0{}); pp_wait_vm<RT0>(); } else pp_wait_vm<0>();
      }
      PP_BARRIER();
      if constexpr (!FIRST) PP_BARRIER();                 // row 1 runs one segment behind
      NT_STAMP(1);
      for (int t = 0; t < nk; ++t) {
        const int s_ = t & 1;
        const uint32_t sb = lds0 + (uint32_t)(s_ * STAGE);
        NT_ABL_STEP(t);
        if constexpr (FIRST) {
          if (t + 1 < nk) { dma(t + 1, s_ ^ 1, A0_BYTES + A1_BYTES, 0, C4{}); dma(t + 1, s_ ^ 1, A0_BYTES + A1_BYTES + PP_BHALF, 4, C4{}); }
          reads(sb);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          PP_BARRIER();
          mfmas();
          PP_FENCE();
          pp_wait_vm<0>();
          PP_BARRIER();
        } else {
          int mode = 0;                                    // what this segment requested: 2 = A1(t+1) and A0(t+2), 1 = A1(t+1), 0 = nothing
          if (t + 2 < nk) { dma(t + 1, s_ ^ 1, A0_BYTES, RT0, CR1{}); dma(t + 2, s_, 0, 0, CR0{}); mode = 2; }
          else if (t + 1 < nk) { dma(t + 1, s_ ^ 1, A0_BYTES, RT0, CR1{}); mode = 1; }
          reads(sb);
          PP_FENCE();
          // A0(t+1), requested one fetch segment ago, has landed: everything but this segment's requests
          if (mode == 2) pp_wait_vm<RT0 + RT1>(); else if (mode == 1) pp_wait_vm<RT1>(); else pp_wait_vm<0>();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          PP_BARRIER();
          mfmas();
          PP_FENCE();
          if (mode == 2) pp_wait_vm<RT0>(); else pp_wait_vm<0>();      // A1(t+1) has landed; A0(t+2) may still be in flight
          PP_BARRIER();
        }
      }
    }
    if constexpr (FIRST) PP_BARRIER();                  // row 1's last multiply segment: afterwards nobody reads operand tiles
    NT_STAMP(2);
#undef PP_BARRIER
#undef PP_FENCE
    const int mw0 = m0 + (FIRST ? 0 : RT0 * 32), nw0 = n0 + wc * 64;
    if constexpr (EpiTraits<VAR>::out_f32 && VAR <= 6) nt_epilogue_f4<2 * RT, VAR>(p, acc, smem, wave, lane, mw0, nw0);
    else nt_epilogue<2 * RT, VAR>(p, acc, smem, wave, lane, mw0, nw0);
#ifdef X2_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NT_STAMP(3);
#endif
  };
  if (wr == 0) run(std::integral_constant<int, RT0>{}, std::true_type{});
  else run(std::integral_constant<int, RT1>{}, std::false_type{});
}

// knobs for A/B measurements and for the tests that pin a kernel variant (0 = automatic everywhere)
//   [0] GROUP_M of the NT tile raster            [1] 0: automatic, 1: 128-column kernels only, 3: always the 256-column kernel
//   [2] NT ablation bits (4 no epilogue, 16 sc1 stores): -DX2_PROBE builds only, refused by the shipped library
//   [6] 1: always the generic (run-time flags) NT epilogue
//   [3] NT tile: 1 = 128x128, 2 = 192x128, 3 = 64x128, 4 = 160x128; 5..8 = rows / 32 of the 256-column kernel        [7] 1: no 160x128 NT tiles
//   [5] TN: 1 = always 128x128 tiles, 2 = 256x256 tiles whenever the contraction lengths allow
//   [9] percent of perfect CU fill the 256-column NT kernel's plan must reach to be chosen automatically (0 = 80)
//   [12] compute units every tile plan leaves out (0 = none): with more than one rank RCCL's channel kernels are resident on
//        some CUs during the backward, and a plan that fills "whole rounds of the 256 CUs" becomes two rounds when a few of
//        them are taken - graph.SegmentedStep sets it to the channel count it caps RCCL at (X2_RESERVED_CUS)
static int g_tune[8] = {0, 0, 0, 0, 0, 0, 0, 0};
//   [10] 256-column NT kernel at 160 rows: 0 = three-stage operand ring with pipelined fragment reads (gemm_nt256s3_kernel<V, true>),
//        2 = three-stage ring, reads at the head of their phase, 1 = the two-stage kernel
//   [11] 1: the rounds 3-4 rule for choosing the 256-column kernel (no fp32 + residual launches below K = 2048, no GELU launches)
//   [13] LayerNorm forward rows per wave (rowwise.hip): 0 automatic, 1 / 2 / 4
//   [14] 1: attention backward always as the dQ + dK/dV pair (attention.hip: no one-pass kernel at 64 < L <= 208)
//   [15] ping-pong NT kernel (gemm_nt_pp_kernel): 0 automatic, 1 never, 3 .. 6 always at 96 / 128 / 160 / 192 rows
static int g_tune_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // keys 8.. : [1] = key 9, [4] = key 12
extern "C" int x2_device_cus(void);
// compute units of the current device (256 on MI355X), asked once: grid-fill decisions below are made in units of it
static int x2_cus() {
  static int n = 0;
  if (n <= 0) { n = x2_device_cus(); if (n <= 0) n = 256; }
  const int left = n - g_tune_x[4];
  return left >= 16 ? left : 16;
}
extern "C" int x2_tune(int key, int value) {
  if (key >= 9 && key <= 15) {
    X2_REQUIRE(key != 12 || value >= 0, "x2_tune: reserved compute units = %d", value);
    g_tune_x[key - 8] = value;
    return X2_OK;
  }
  X2_REQUIRE(key >= 0 && key < 8, "x2_tune: no key %d", key);
#ifndef X2_PROBE
  X2_REQUIRE(key != 2 || value == 0, "x2_tune: NT ablation bits (key 2, value %d) exist in -DX2_PROBE builds only", value);
#endif
  g_tune[key] = value;
  return X2_OK;
}
#ifdef X2_PROBE
extern "C" int x2_probe_set_buffer(void* buf) {      // probe builds only: not in include/x2vlm_hip.h
  unsigned long long* b = (unsigned long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_nt_probe), &b, sizeof(b)) == hipSuccess ? X2_OK : X2_ERR_LAUNCH;
}
#endif
// current value of a knob (bench.py reports every non-default one in its JSON line); -1 for a key that does not exist
extern "C" int x2_tune_get(int key) {
  if (key >= 9 && key <= 15) return g_tune_x[key - 8];
  if (key < 0 || key >= 8) return -1;
  return g_tune[key];
}

// 256-column kernel: tile height for this problem.  A launch is a few rounds of one workgroup per CU, each round as long as
// its tiles are high: cost = rounds x TMW; efficiency = (M x N work spread perfectly) / cost.  Returns TMW (5..8) of the best
// height and its efficiency in *eff.
static int nt256_plan(int M, int N, double* eff) {
  const int cus = x2_cus(), tiles_n = (N + 255) / 256;
  const double ideal = (double)M * N / (512.0 * 256.0 * cus);            // in units of one 32-row-per-TMW tile slice
  int best = 8; double best_cost = 1e30;
  for (int t = 8; t >= 5; --t) {
    const long tiles = (long)((M + 32 * t - 1) / (32 * t)) * tiles_n;
    const double cost = (double)((tiles + cus - 1) / cus) * t / 16.0;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = t; }
  }
  if (eff) *eff = ideal / best_cost;
  return best;
}
template <int TMW, int V>
static void launch_nt256(const GemmNT& p, hipStream_t stream) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256_kernel<TMW, V>), hipFuncAttributeMaxDynamicSharedMemorySize, N2_LDS_BYTES);
    raised = true;
  }
  const int tiles = ((p.M + 32 * TMW - 1) / (32 * TMW)) * ((p.N + 255) / 256);
  hipLaunchKernelGGL((gemm_nt256_kernel<TMW, V>), dim3(tiles), dim3(512), N2_LDS_BYTES, stream, p);
}
template <int V>
static void launch_nt256s3(const GemmNT& p, hipStream_t stream) {
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256s3_kernel<V, true>), hipFuncAttributeMaxDynamicSharedMemorySize, N3_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt256s3_kernel<V, false>), hipFuncAttributeMaxDynamicSharedMemorySize, N3_LDS_BYTES);
    raised = true;
  }
  const int tiles = ((p.M + 159) / 160) * ((p.N + 255) / 256);
  if (g_tune_x[2] == 2) hipLaunchKernelGGL((gemm_nt256s3_kernel<V, false>), dim3(tiles), dim3(512), N3_LDS_BYTES, stream, p);
  else hipLaunchKernelGGL((gemm_nt256s3_kernel<V, true>), dim3(tiles), dim3(512), N3_LDS_BYTES, stream, p);
}
template <int V>
static void launch_nt256_h(const GemmNT& p, int tmw, hipStream_t stream) {
  switch (tmw) {
    case 5: if (g_tune_x[2] == 1) launch_nt256<5, V>(p, stream); else launch_nt256s3<V>(p, stream); break;
    case 6: launch_nt256<6, V>(p, stream); break;
    case 7: launch_nt256<7, V>(p, stream); break;
    default: launch_nt256<8, V>(p, stream); break;
  }
}

// ping-pong kernel: height h = 4 / 5 / 6 32-row tiles (128 / 160 / 192 rows), split over the two wave rows as 2+2 / 2+3 / 3+3
template <int RT0, int RT1, int V, int NST>
static void launch_nt_pp(const GemmNT& p, hipStream_t stream) {
  constexpr int lds = NST * ((RT0 + RT1) * 4096 + 2 * PP_BHALF);
  static bool raised = false;
  if (!raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<RT0, RT1, V, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    raised = true;
  }
  constexpr int BMT = 32 * (RT0 + RT1);
  const int tiles = ((p.M + BMT - 1) / BMT) * ((p.N + 255) / 256);
  hipLaunchKernelGGL((gemm_nt_pp_kernel<RT0, RT1, V, NST>), dim3(tiles), dim3(512), lds, stream, p);
}
template <int V>
static void launch_nt_pp_h(const GemmNT& p, int h, hipStream_t stream) {
  switch (h) {
    case 3: launch_nt_pp<1, 2, V, 3>(p, stream); break;
    case 4: if (g_tune_x[2] == 1) launch_nt_pp<2, 2, V, 2>(p, stream); else launch_nt_pp<2, 2, V, 3>(p, stream); break;
    case 5: if (g_tune_x[2] == 1) launch_nt_pp<2, 3, V, 2>(p, stream); else launch_nt_pp<2, 3, V, 3>(p, stream); break;
    default: launch_nt_pp<3, 3, V, 2>(p, stream); break;
  }
}

// one NT launch: LDS = two stages of a (32 * TM) x 64 A tile + a 128 x 64 B tile; above 64 KB the limit is raised once
template <int TM, int V, bool F4>
static void launch_nt(const GemmNT& p, int tiles, hipStream_t stream) {
  constexpr int lds = 2 * (32 * TM * 128 + TILE_BYTES);
  static bool raised = false;
  if (lds > 65536 && !raised) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_kernel<TM, V, F4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    raised = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<TM, V, F4>), dim3(tiles), dim3(256), lds, stream, p);
}

extern "C" int x2_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                          const float* bias, const float* gamma, const float* resid, int ldr, void* aux, int ldaux,
                          int act, int out_f32, unsigned drop_thr16, unsigned drop_seed, float drop_scale, const unsigned* drop_epoch,
                          const float* rowscale, float* colsum, void* stream) {
  X2_REQUIRE(M > 0 && N > 0 && K > 0, "x2_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  X2_REQUIRE(drop_thr16 < 65536u, "x2_gemm_nt: drop_thr16=%u", drop_thr16);
  X2_REQUIRE(K % BK == 0, "x2_gemm_nt: K=%d must be a multiple of %d", K, BK);
  X2_REQUIRE(N % 8 == 0, "x2_gemm_nt: N=%d must be a multiple of 8", N);
  X2_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "x2_gemm_nt: leading dims must keep 16-byte rows");
  X2_REQUIRE(act == 0 || aux, "x2_gemm_nt: act=%d needs aux", act);
  X2_REQUIRE((!resid || ldr % 4 == 0) && (!aux || ldaux % 8 == 0), "x2_gemm_nt: ldr/ldaux alignment");
  GemmNT p{(const bf16_t*)A, (const bf16_t*)B, C, bias, gamma, resid, (bf16_t*)aux, M, N, K, lda, ldb, ldc, ldr, ldaux, act, out_f32,
           g_tune[0] > 0 ? g_tune[0] : 8, DropSpec{drop_thr16, drop_seed, drop_scale}, drop_epoch, rowscale, colsum, g_tune[2], 0};
  // epilogue variant (EpiTraits): the feature sets the step uses are compiled separately, anything else runs the generic one
  int var = 4;
  {
    const bool plain = !gamma && !rowscale && !drop_thr16 && !colsum;
    if (g_tune[6] == 1) var = 4;                                                     // probes / tests: force the generic epilogue
    else if (act == 1 && !out_f32 && !resid && plain) var = 2;
    else if (act == 2 && !out_f32 && !resid && plain) var = 3;
    else if (act == 0 && !aux && !resid && plain) var = out_f32 ? 1 : 0;
    else if (act == 0 && !aux && resid && out_f32 && !gamma && !rowscale && !colsum) var = 5;
    else if (act == 0 && !aux && resid && out_f32 && gamma && !drop_thr16 && !colsum) var = 6;
  }
  // 256-column kernel (gemm_nt256_kernel): [1] = 3 always (tile height from [3] = 5..8 or the plan), [1] = 1 never; the
  // generic feature set stays on the kernels above
  double eff256 = 0.0;
  int tmw = nt256_plan(M, N, &eff256);
  if (g_tune[3] >= 5 && g_tune[3] <= 8) tmw = g_tune[3];
  // Automatic choice ([1] = 0), from the per-shape A/B of probes/bench_nt256.py (profiles/r03b_nt256_per_shape.txt): with one
  // workgroup per CU nothing covers a tile's load prologue and its epilogue, so the 256-column kernel wins where those are
  // a small part of the tile - long contractions (K >= 2048: 1.19 vs 1.07 PFLOP/s main loops) or single-output bf16 / fp32
  // epilogues - and its plan fills >= 80 % of the CU rounds; never for the short fp32 + residual launches of the text rows.
  const bool light = var == 0 || var == 1;
  bool auto256 = (light || K >= 2048) && eff256 * 100.0 >= (g_tune_x[1] > 0 ? g_tune_x[1] : 80) && !(resid && M < 8192);
  // Round 5, with the three-stage ring at 160 rows and outputs that are NOT cache-resident (probes/bench_nt_choice.py,
  // profiles/r09k_nt_kernel_choice.txt): the fp32 + residual launches of the long-row towers win on this kernel at any K (vision proj
  // 37.2 -> 28.8 us, X2VLM-large 68.1 -> 58.5, the image-token input gradient of the cross-attentions 41.4 -> 40.1), and so do the
  // GELU launches whose plan fills >= 90 % of its rounds (vision fc1 92.6 -> 88.4, fusion ffn1 57.6 -> 55.1 at 192 rows, X2VLM-large
  // fc1 199.6 -> 183.7 at 256 rows); the text / fusion rows (M < 8192) with a residual stay where they were.  x2_tune(11, 1): the old rule.
  if (g_tune_x[3] != 1) {
    if ((var == 5 || var == 6) && M >= 8192 && eff256 * 100.0 >= (g_tune_x[1] > 0 ? g_tune_x[1] : 80)) auto256 = true;
    if (var == 2 && eff256 * 100.0 >= 90.0) auto256 = true;
  }
  const bool use256 = var != 4 && N % 8 == 0 && (g_tune[1] == 3 || ((g_tune[1] == 0 || g_tune[1] == 4) && g_tune[3] == 0 && auto256));
  // Ping-pong kernel (gemm_nt_pp_kernel).  x2_tune(15, h), h = 3 .. 6: always, at 32 h rows (probes); 1: never; 0: automatic - the launches the 128-column kernels
  // serve worst: a long contraction (K >= 2048) over few output tiles (the fusion stack's ffn2 forward and ffn1 input gradient, M = 7680 x N = 768: 180 tiles of
  // 128 x 128, 48 serial steps each, 54-61 us), where ONE round of 96- or 128-row x 256-column tiles fills >= 70 % of the CUs: 44-50 us (profiles/r12a_nt_pp_tail.txt).
  // Same sums in the same order as every other NT kernel (bit-identical results: probes/bench_nt_pp.py).  Operands must be addressable through a 2 GB descriptor.
  const bool pp_ok = var != 4 && ((size_t)(M - 1) * lda + K) * 2 < (1ull << 31) && ((size_t)(N - 1) * ldb + K) * 2 < (1ull << 31);
  int pp_h = 0;
  if (g_tune_x[7] >= 3 && g_tune_x[7] <= 6) pp_h = g_tune_x[7];
  else if (g_tune_x[7] == 0 && g_tune[1] == 0 && g_tune[3] == 0 && !use256 && K >= 2048 && (var == 0 || var == 1 || var == 5)) {
    const int cus = x2_cus(), tn256 = (N + 255) / 256;
    for (int h = 3; h <= 4 && !pp_h; ++h) {
      const int tiles = ((M + 32 * h - 1) / (32 * h)) * tn256;
      if (tiles <= cus && tiles * 10 >= cus * 7) pp_h = h;
    }
  }
  if (pp_ok && pp_h) {
    switch (var) {
      case 0: launch_nt_pp_h<0>(p, pp_h, (hipStream_t)stream); break;
      case 1: launch_nt_pp_h<1>(p, pp_h, (hipStream_t)stream); break;
      case 2: launch_nt_pp_h<2>(p, pp_h, (hipStream_t)stream); break;
      case 3: launch_nt_pp_h<3>(p, pp_h, (hipStream_t)stream); break;
      case 5: launch_nt_pp_h<5>(p, pp_h, (hipStream_t)stream); break;
      default: launch_nt_pp_h<6>(p, pp_h, (hipStream_t)stream); break;
    }
    return x2_check_launch("x2_gemm_nt");
  }
  if (use256) {
    switch (var) {
      case 0: launch_nt256_h<0>(p, tmw, (hipStream_t)stream); break;
      case 1: launch_nt256_h<1>(p, tmw, (hipStream_t)stream); break;
      case 2: launch_nt256_h<2>(p, tmw, (hipStream_t)stream); break;
      case 3: launch_nt256_h<3>(p, tmw, (hipStream_t)stream); break;
      case 5: launch_nt256_h<5>(p, tmw, (hipStream_t)stream); break;
      default: launch_nt256_h<6>(p, tmw, (hipStream_t)stream); break;
    }
  } else {
    // 192x128 tiles when that gives a single resident round (<= 2 workgroups per CU) where 128x128 needs a second,
    // nearly empty one: the N = 768 outputs of this model (66 x 6 = 396 tiles vs 99 x 6 = 594 on 512 slots).
    // 64x128 tiles when 128x128 would leave CUs with a single (or no) workgroup: the text / fusion rows (M = 3840, 7680)
    // times N = 768 give 180 / 360 tiles, each a serial chain of K/64 load->wait->multiply steps with nothing else on the
    // CU to hide the load latency; half-height tiles double the chains in flight (3 fit a CU: 48 KB LDS each).
    const int t128 = ((M + 127) / 128) * ((N + BN - 1) / BN), t192 = ((M + 191) / 192) * ((N + BN - 1) / BN);
    const int t64 = ((M + 63) / 64) * ((N + BN - 1) / BN);
    // measured (probes/bench_gemm.py): 64x128 wins up to ~1.2 workgroups of 128x128 per CU slot pair, except on long
    // contractions with exactly one 192x128 round (fc2, dqkv: 757 / 817 TFLOP/s on 192x128)
    const int slots = 2 * x2_cus();                      // two 128x128 workgroups per CU
    const bool use64 = g_tune[3] == 3 || (g_tune[3] == 0 && (t128 <= slots || (t128 <= slots + slots / 6 && K <= 1024)));
    // (192x128 on the long-row shapes of X2VLM-large - qkv / fc1 at K = 1024, fc2 at K = 4096 - is 4-9 % faster in
    // probes/bench_gemm_large.py but neutral inside the step: not selected)
    const bool use192 = !use64 && (g_tune[3] == 2 || (g_tune[3] == 0 && t128 > slots && t192 <= slots));
    // 160x128 where 192x128 was chosen to save a round and 160-row tiles still fit that one round: the vision N = 768
    // outputs are 79 x 6 = 474 tiles instead of 66 x 6 = 396, i.e. the busiest CU holds 2 x 160 rows instead of 2 x 192
    // (balanced would be 148): x2_tune(7, 1) switches the rule off
    const int t160 = ((M + 159) / 160) * ((N + BN - 1) / BN);
    const bool use160 = g_tune[3] == 4 || (use192 && g_tune[3] == 0 && g_tune[7] != 1 && t160 <= slots);
    // fp32-out feature sets (1, 5, 6) store row-contiguously (nt_epilogue_f4: in-step A/B 23.33-23.38 vs 23.42-23.43 ms per base step,
    // profiles/r05i_knob_ab.txt - the 8-columns-per-lane form of these three sets was removed in round 4)
#define X2_NT_LAUNCH(V, SW)                                                                          \
    do {                                                                                              \
      if (use64) launch_nt<2, V, SW>(p, t64, (hipStream_t)stream);                                    \
      else if (use160) launch_nt<5, V, SW>(p, t160, (hipStream_t)stream);                             \
      else if (use192) launch_nt<6, V, SW>(p, t192, (hipStream_t)stream);                             \
      else launch_nt<4, V, SW>(p, t128, (hipStream_t)stream);                                         \
    } while (0)
    switch (var) {
      case 0: X2_NT_LAUNCH(0, false); break;
      case 2: X2_NT_LAUNCH(2, false); break;
      case 3: X2_NT_LAUNCH(3, false); break;
      case 1: X2_NT_LAUNCH(1, true); break;
      case 5: X2_NT_LAUNCH(5, true); break;
      case 6: X2_NT_LAUNCH(6, true); break;
      default: X2_NT_LAUNCH(4, false); break;
    }
#undef X2_NT_LAUNCH
  }
  return x2_check_launch("x2_gemm_nt");
}

// C = (A . B^T) x GELU'(aux) -> bf16 (epilogue variant 10) plus the column sums of C as partial rows: colparts[r][N], r <
// *nrows_out = 2 x row tiles of the launch (a row past the last wave row that holds data is all zeros); the caller adds
// them up with x2_reduce_partials(_multi)(colparts, *nrows_out, 1, N, out).  colparts must hold 2 * ceil(M / 64) rows.
// Input gradient through the GELU of the MLPs + the bias gradient of fc1 / intermediate.dense (beit2.py:62-66, xbert.py:497).
extern "C" int x2_gemm_nt_dgelu_colparts(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                         const void* aux, int ldaux, float* colparts, int* nrows_out, void* stream) {
  X2_REQUIRE(A && B && C && aux && colparts && nrows_out, "x2_gemm_nt_dgelu_colparts: null argument");
  X2_REQUIRE(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 8 == 0, "x2_gemm_nt_dgelu_colparts: M=%d N=%d K=%d", M, N, K);
  X2_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldaux % 8 == 0, "x2_gemm_nt_dgelu_colparts: leading dims must keep 16-byte rows");
  GemmNT p{(const bf16_t*)A, (const bf16_t*)B, C, nullptr, nullptr, nullptr, (bf16_t*)aux, M, N, K, lda, ldb, ldc, 0, ldaux, 2, 0,
           g_tune[0] > 0 ? g_tune[0] : 8, DropSpec{0u, 0u, 1.f}, nullptr, nullptr, colparts, 0, 0};
  // the tile rule of x2_gemm_nt for this feature set (never the 160-row tile: it exists to balance N = 768 outputs)
  const int tiles_n = (N + BN - 1) / BN, t128 = ((M + 127) / 128) * tiles_n, t192 = ((M + 191) / 192) * tiles_n, t64 = ((M + 63) / 64) * tiles_n;
  const int slots = 2 * x2_cus();
  const bool use64 = g_tune[3] == 3 || (g_tune[3] == 0 && (t128 <= slots || (t128 <= slots + slots / 6 && K <= 1024)));
  const bool use192 = !use64 && (g_tune[3] == 2 || (g_tune[3] == 0 && t128 > slots && t192 <= slots));
  if (use64) { launch_nt<2, 10, false>(p, t64, (hipStream_t)stream); *nrows_out = 2 * ((M + 63) / 64); }
  else if (use192) { launch_nt<6, 10, false>(p, t192, (hipStream_t)stream); *nrows_out = 2 * ((M + 191) / 192); }
  else { launch_nt<4, 10, false>(p, t128, (hipStream_t)stream); *nrows_out = 2 * ((M + 127) / 128); }
  return x2_check_launch("x2_gemm_nt_dgelu_colparts");
}

// ---------------------------------------------------------------------------------------------
// NT with a split contraction: C[M,N] (fp32, no epilogue) = A[M,K] . B[N,K]^T for few output tiles and a long K - the
// input gradient of the tied MLM decoder, dt[R,768] = dlogits[R,30528] . E^T, is 36-72 tiles of 477 contraction steps,
// i.e. a 354 us serial chain on a seventh of the chip.  Slice s of the contraction (gridDim.y) writes its M x N partial
// product to ws[s] through the ordinary NT kernel (epilogue variant 7), gemm_nt_sum_slices_kernel adds the slices in a
// fixed order (deterministic, no atomics).  slices = 0: as many as fill two workgroups per CU, at least 8 steps each.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_nt_sum_slices_kernel(const float* __restrict__ ws, float* __restrict__ C, int M, int N, int ldc, int S) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x, per = (long)M * N / 4;      // one float4 of the M x N result
  if (e >= per) return;
  const int m = (int)(e * 4 / N), n = (int)(e * 4 % N);
  float4 o = *reinterpret_cast<const float4*>(ws + e * 4);
  for (int s_ = 1; s_ < S; ++s_) {
    const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)s_ * M * N + e * 4);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
  }
  *reinterpret_cast<float4*>(C + (size_t)m * ldc + n) = o;
}
extern "C" int x2_gemm_nt_splitk(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                 int slices, float* ws, long ws_floats, void* stream) {
  X2_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "x2_gemm_nt_splitk: empty problem M=%d N=%d K=%d", M, N, K);
  X2_REQUIRE(K % BK == 0 && N % 8 == 0, "x2_gemm_nt_splitk: K=%d must be a multiple of %d, N=%d of 8", K, BK, N);
  X2_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && ldc >= N, "x2_gemm_nt_splitk: leading dims must keep 16-byte rows");
  X2_REQUIRE(slices >= 0, "x2_gemm_nt_splitk: slices=%d", slices);
  const int nk = K / BK, tiles_n = (N + BN - 1) / BN;
  const bool small = M <= 64;                                     // 64-row tiles when there is a single short row panel
  const int tiles = (small ? (M + 63) / 64 : (M + 127) / 128) * tiles_n;
  int S = slices;
  if (S == 0) { S = (2 * x2_cus() + tiles - 1) / tiles; if (S > nk / 8) S = nk / 8; }
  if (S > nk) S = nk;
  const long per = (long)M * N;
  if (ws == nullptr) S = 1;
  else if ((long)S * per > ws_floats) S = (int)(ws_floats / per);
  if (S < 1) S = 1;
  const int steps = (nk + S - 1) / S;                             // contraction steps per slice; the last slice may be shorter
  S = (nk + steps - 1) / steps;                                   // no empty slice
  GemmNT p{(const bf16_t*)A, (const bf16_t*)B, S > 1 ? (void*)ws : (void*)C, nullptr, nullptr, nullptr, nullptr, M, N, K, lda, ldb,
           S > 1 ? N : ldc, 0, 0, 0, 1, g_tune[0] > 0 ? g_tune[0] : 8, DropSpec{0u, 0u, 1.f}, nullptr, nullptr, nullptr, 0, steps * BK};
  if (small) hipLaunchKernelGGL((gemm_nt_kernel<2, 7>), dim3(tiles, S), dim3(256), 2 * (64 * 128 + TILE_BYTES), (hipStream_t)stream, p);
  else hipLaunchKernelGGL((gemm_nt_kernel<4, 7>), dim3(tiles, S), dim3(256), GEMM_LDS_BYTES, (hipStream_t)stream, p);
  if (S > 1)
    hipLaunchKernelGGL(gemm_nt_sum_slices_kernel, dim3((unsigned)((per / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, C, M, N, ldc, S);
  return x2_check_launch("x2_gemm_nt_splitk");
}

// ---------------------------------------------------------------------------------------------
// MLM head: cross-entropy over the tied decoder WITHOUT materialising the [R, Vp] fp32 logits (94 MB written once and
// read twice per step at R = 768).  Forward: the decoder GEMM's epilogue reduces every 64-column chunk of a row to
// (max, sum exp) and picks the logit at the label; x2_ce_combine (heads.hip) folds the Vp / 64 chunks of a row into its
// log-partition and the mean loss.  Backward: the same GEMM again (36 GFLOP, ~45 us) with an epilogue that turns the
// accumulators straight into (softmax - onehot) * g / count in bf16 - the operand of the two gradient GEMMs that follow.
// Reference: BertLMPredictionHead.decoder + CrossEntropyLoss, xbert.py:822, 1653-1661.
// ---------------------------------------------------------------------------------------------
static int launch_nt_ce(GemmNT& p, int var, hipStream_t stream) {
  const int tiles_n = (p.N + BN - 1) / BN, t128 = ((p.M + 127) / 128) * tiles_n, t64 = ((p.M + 63) / 64) * tiles_n;
  const bool use64 = t128 <= 2 * x2_cus();
  if (var == 8) {
    if (use64) hipLaunchKernelGGL((gemm_nt_kernel<2, 8>), dim3(t64), dim3(256), 2 * (64 * 128 + TILE_BYTES), stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<4, 8>), dim3(t128), dim3(256), GEMM_LDS_BYTES, stream, p);
  } else {
    if (use64) hipLaunchKernelGGL((gemm_nt_kernel<2, 9>), dim3(t64), dim3(256), 2 * (64 * 128 + TILE_BYTES), stream, p);
    else hipLaunchKernelGGL((gemm_nt_kernel<4, 9>), dim3(t128), dim3(256), GEMM_LDS_BYTES, stream, p);
  }
  return 0;
}
static bool mlm_ce_shapes_ok(int R, int Vp, int V, int Hd, int ldx, int lde) {
  return R > 0 && V > 0 && V <= Vp && Vp % 64 == 0 && Hd > 0 && Hd % BK == 0 && ldx % 8 == 0 && lde % 8 == 0 && ldx >= Hd && lde >= Hd;
}
extern "C" int x2_mlm_ce_fwd(const void* X, const void* E, const float* bias, const long* labels, int R, int Vp, int V, int Hd,
                             int ldx, int lde, float* part, float* zlab, void* stream) {
  X2_REQUIRE(X && E && labels && part && zlab, "x2_mlm_ce_fwd: null argument");
  X2_REQUIRE(mlm_ce_shapes_ok(R, Vp, V, Hd, ldx, lde), "x2_mlm_ce_fwd: R=%d Vp=%d V=%d Hd=%d ldx=%d lde=%d", R, Vp, V, Hd, ldx, lde);
  GemmNT p{(const bf16_t*)X, (const bf16_t*)E, nullptr, bias, nullptr, nullptr, nullptr, R, Vp, Hd, ldx, lde, Vp, 0, 0, 0, 0,
           g_tune[0] > 0 ? g_tune[0] : 8, DropSpec{0u, 0u, 1.f}, nullptr, nullptr, nullptr, 0, 0,
           labels, nullptr, nullptr, nullptr, part, zlab, 1.f, V};
  launch_nt_ce(p, 8, (hipStream_t)stream);
  return x2_check_launch("x2_mlm_ce_fwd");
}
extern "C" int x2_mlm_ce_bwd(const void* X, const void* E, const float* bias, const long* labels, const float* lse, const float* g,
                             const float* stat, float gscale, int R, int Vp, int V, int Hd, int ldx, int lde, void* dl_bf16, long ldd,
                             void* stream) {
  X2_REQUIRE(X && E && labels && lse && g && stat && dl_bf16, "x2_mlm_ce_bwd: null argument");
  X2_REQUIRE(mlm_ce_shapes_ok(R, Vp, V, Hd, ldx, lde) && ldd >= Vp && ldd % 8 == 0, "x2_mlm_ce_bwd: R=%d Vp=%d V=%d Hd=%d ldd=%ld", R, Vp, V, Hd, ldd);
  GemmNT p{(const bf16_t*)X, (const bf16_t*)E, dl_bf16, bias, nullptr, nullptr, nullptr, R, Vp, Hd, ldx, lde, (int)ldd, 0, 0, 0, 0,
           g_tune[0] > 0 ? g_tune[0] : 8, DropSpec{0u, 0u, 1.f}, nullptr, nullptr, nullptr, 0, 0,
           labels, lse, g, stat, nullptr, nullptr, gscale, V};
  launch_nt_ce(p, 9, (hipStream_t)stream);
  return x2_check_launch("x2_mlm_ce_bwd");
}

// ---------------------------------------------------------------------------------------------
// TN (weight gradients): C[n][k] (+)= sum_m A[m][n] * B[m][k].  Operand tiles are [64 m][128 cols]
// (row-major as they sit in HBM, 256-byte rows), 16 chunks of 16 B per row, chunk c of row r at
// position c ^ (f(r) << 1), f(r) = (r & 3) | ((r >> 3 & 1) << 2)  -> conflict-free transposing reads.
// Grouped: up to 8 problems per launch (all weight gradients of one layer), optional split over
// the contraction (gridDim.y) with fp32 atomics when the layer has too few tiles to fill 256 CUs.
// ---------------------------------------------------------------------------------------------
struct TNProblem {
  const bf16_t* A; const bf16_t* B; float* C;
  int Mc, N, K;           // contraction length, output rows, output cols
  int lda, ldb, ldc;
  int n_ld, k_ld;         // readable columns of A / B rows (>= N / K, multiple of 8)
  int tile_begin, tiles_k;
};
struct GemmTNGroup { TNProblem p[8]; int count; int accumulate; };

__device__ __forceinline__ int tn_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTNGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  int t = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < g.count && t >= g.p[i].tile_begin) pi = i;
  const TNProblem p = g.p[pi];
  t -= p.tile_begin;
  int tnn, tkk;
  tile_coords(t, (p.N + 127) / 128, p.tiles_k, 8, tnn, tkk);
  const int n0 = tnn * 128, k0 = tkk * 128;
  const int steps = (p.Mc + BK - 1) / BK;
  const int per = (steps + gridDim.y - 1) / gridDim.y;
  const int s_begin = blockIdx.y * per, s_end = min(steps, s_begin + per);
  if (s_begin >= s_end) return;

  const bf16_t* srcA[4]; const bf16_t* srcB[4]; int rowq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = i * 256 + tid, row = q >> 4, c = (q & 15) ^ (tn_f(row) << 1);
    int ca = n0 + c * 8; ca = ca <= p.n_ld - 8 ? ca : p.n_ld - 8;
    int cb = k0 + c * 8; cb = cb <= p.k_ld - 8 ? cb : p.k_ld - 8;
    rowq[i] = row;
    srcA[i] = p.A + (size_t)row * p.lda + ca;
    srcB[i] = p.B + (size_t)row * p.ldb + cb;
  }
  auto stage = [&](int s, int buf) {
    char* base = smem + buf * STAGE_BYTES;
    const int mbase = s * BK;
    if (mbase + BK <= p.Mc) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        glds16(srcA[i] + (size_t)mbase * p.lda, base + (i * 256 + wave * 64) * 16);
        glds16(srcB[i] + (size_t)mbase * p.ldb, base + TILE_BYTES + (i * 256 + wave * 64) * 16);
      }
    } else {  // ragged last step: rows past Mc contribute zeros
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 va = u32x4{0, 0, 0, 0}, vb = u32x4{0, 0, 0, 0};
        if (mbase + rowq[i] < p.Mc) {
          va = *reinterpret_cast<const u32x4*>(srcA[i] + (size_t)mbase * p.lda);
          vb = *reinterpret_cast<const u32x4*>(srcB[i] + (size_t)mbase * p.ldb);
        }
        *reinterpret_cast<u32x4*>(base + (i * 256 + tid) * 16) = va;
        *reinterpret_cast<u32x4*>(base + TILE_BYTES + (i * 256 + tid) * 16) = vb;
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = lds_addr(smem);
  const int fi = lane & 15, fg = lane >> 4, fr = fi >> 2, fc4 = fi & 3;
  const int fsw = (fr | ((fg & 1) << 2)) << 1;                 // f(R) << 1 for every row this lane addresses
  // byte offset of this lane's address inside a tile for MFMA k-step ks, second half adds 4 rows
  const uint32_t rowoff = (uint32_t)((fg * 8 + fr) * 256 + (fc4 & 1) * 8);

  stage(s_begin, 0);
  for (int s = s_begin; s < s_end; ++s) {
    const int it = s - s_begin;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < s_end) stage(s + 1, (it + 1) & 1);
    const uint32_t sb = lds0 + (it & 1) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xa[4], ya[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {   // B operand tile (activations X): output k index
        const uint32_t ch = (uint32_t)(((2 * (wk * 4 + i) + (fc4 >> 1)) ^ fsw) << 4);
        const uint32_t a0 = sb + TILE_BYTES + ks * 32 * 256 + rowoff + ch;
        xa[i] = lds_read_tr_frag(a0, a0 + 4 * 256);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // A operand tile (output grads dY): output n index
        const uint32_t ch = (uint32_t)(((2 * (wn * 4 + j) + (fc4 >> 1)) ^ fsw) << 4);
        const uint32_t a0 = sb + ks * 32 * 256 + rowoff + ch;
        ya[j] = lds_read_tr_frag(a0, a0 + 4 * 256);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // D[row = k][col = n]: lane holds n = fi and 4 consecutive k -> 16-byte stores along K
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i], ya[j], acc[i][j], 0, 0, 0);
    }
  }

  const bool atomic = gridDim.y > 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wn * 64 + j * 16 + fi;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + wk * 64 + i * 16 + fg * 4;
      if (k >= p.K) continue;
      float* dst = p.C + (size_t)n * p.ldc + k;
      if (atomic) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(dst + r, acc[i][j][r]);
      } else if (g.accumulate) {
        float4 o = *reinterpret_cast<float4*>(dst);
        o.x += acc[i][j][0]; o.y += acc[i][j][1]; o.z += acc[i][j][2]; o.w += acc[i][j][3];
        *reinterpret_cast<float4*>(dst) = o;
      } else {
        *reinterpret_cast<float4*>(dst) = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// TN, large tile: 256(n) x 256(k) outputs per 512-thread workgroup (8 waves as 2 x 4, 128 x 64 each = 8 x 4 MFMA
// tiles, 128 accumulator VGPRs), one workgroup per CU.  Weight gradients are the one GEMM family of this step with a
// long contraction (Mc = 3840..12608 rows) and a negligible epilogue, so what bounds them is LDS traffic per MFMA:
// a 128x64 wave tile reads 0.375 fragments per MFMA (0.5 at 64x64) and the block stages 1/128 instead of 1/64 bytes
// per flop through the global->LDS path.
//   LDS: 2 contraction steps x 4 half-tiles (A0, A1 = dY columns n0.., n0+128.. ; B0, B1 = X columns) x 16 KB, each a
//   [64 m][128 cols] image in the swizzle of the kernel above (wave wr reads only A[wr], wave wc only B[wc >> 1]).
//   One contraction step = 4 phases of 16 MFMAs per wave (quadrants A-lo x B-lo, A-lo x B-hi, A-hi x B-hi, A-hi x B-lo
//   of the wave tile); every phase also requests ONE half-tile:
//     phase 1, 2: A0, A1 of step t+1 into the other buffer (its last reads were step t-1, behind the end barrier)
//     phase 3, 4: B0, B1 of step t+2 into THIS buffer (all B reads of step t are over after phase 2: mid barrier)
//   so 2-4 half-tiles are always in flight and each has at least two phases (32 MFMAs per wave) to land before the
//   counted s_waitcnt vmcnt(4) + barrier at the end of the step that precedes its first read.
// Split contraction (gridDim.y > 1): partial tiles go to `ws` in fragment order (1 KB per store instruction) and
// gemm_tn256_reduce_kernel adds them in a fixed order: deterministic, no atomics (14 M strided fp32 atomics of a
// split-2 layer cost more than the GEMM itself: probes/bench_tn_split.py).
// ---------------------------------------------------------------------------------------------
#define T2_HALF 16384
#define T2_LDS_BYTES (8 * T2_HALF)
// source of the rows past a ragged contraction length (Mc % 64 != 0: X2VLM-large has 32 x 577 rows): the LDS-DMA lanes of
// those rows read these 16 zero bytes instead, so the last contraction step multiplies zeros - no padded copies of the
// operands, no out-of-bounds reads
__device__ __attribute__((aligned(16))) const uint32_t x2_zero_chunk[4] = {0u, 0u, 0u, 0u};
template <bool RAG>      // RAG: some problem of the launch has a contraction length that is not a multiple of 64
__global__ __launch_bounds__(512) void gemm_tn256_kernel(GemmTNGroup g, float* ws, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < g.count && tile >= g.p[i].tile_begin) pi = i;
  const TNProblem p = g.p[pi];
  int tnn, tkk;
  tile_coords(tile - p.tile_begin, (p.N + 255) / 256, p.tiles_k, 4, tnn, tkk);
  const int n0 = tnn * 256, k0 = tkk * 256;
  const int steps = (p.Mc + BK - 1) / BK, rag = p.Mc % BK;        // host guarantees steps >= gridDim.y; rag = rows of a partial last step
  const int s_begin = (int)((long)steps * blockIdx.y / gridDim.y), s_end = (int)((long)steps * (blockIdx.y + 1) / gridDim.y);
  const int nk = s_end - s_begin;
  const int last_rel = (RAG && rag != 0 && s_end == steps) ? nk - 1 : -1;  // slice-relative index of the partial step, if this slice has it
  const bf16_t* const zsrc = reinterpret_cast<const bf16_t*>(x2_zero_chunk);
  const bool oob0 = (tid >> 4) >= rag, oob1 = ((512 + tid) >> 4) >= rag;   // this thread's two rows of a half-tile

  // per-thread global sources: half-tile h of A / B, chunk i (2 x 512 chunks of 16 B per half-tile)
  const bf16_t* srcA[2][2]; const bf16_t* srcB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = i * 512 + tid, row = q >> 4, c = (q & 15) ^ (tn_f(row) << 1);
      int ca = n0 + h * 128 + c * 8; ca = ca <= p.n_ld - 8 ? ca : p.n_ld - 8;
      int cb = k0 + h * 128 + c * 8; cb = cb <= p.k_ld - 8 ? cb : p.k_ld - 8;
      srcA[h][i] = p.A + ((size_t)s_begin * BK + row) * p.lda + ca;
      srcB[h][i] = p.B + ((size_t)s_begin * BK + row) * p.ldb + cb;
    }
  const size_t stepA = (size_t)BK * p.lda, stepB = (size_t)BK * p.ldb;
  auto issue = [&](int kt, auto slot) {                            // slot 0, 1: A halves; 2, 3: B halves
    constexpr int S = decltype(slot)::value;
    char* base = smem + ((kt & 1) * 4 + S) * T2_HALF + wave * 1024;
    const bool z = RAG && kt == last_rel;
    if constexpr (S < 2) {
      glds16(z && oob0 ? zsrc : srcA[S][0] + kt * stepA, base);
      glds16(z && oob1 ? zsrc : srcA[S][1] + kt * stepA, base + 8192);
    } else {
      glds16(z && oob0 ? zsrc : srcB[S - 2][0] + kt * stepB, base);
      glds16(z && oob1 ? zsrc : srcB[S - 2][1] + kt * stepB, base + 8192);
    }
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = lds_addr(smem);
  const int fi = lane & 15, fg = lane >> 4, fr = fi >> 2, fc4 = fi & 3;
  const int fsw = (fr | ((fg & 1) << 2)) << 1;
  const uint32_t rowoff = (uint32_t)((fg * 8 + fr) * 256 + (fc4 & 1) * 8);
  const uint32_t offA = (uint32_t)(wr * T2_HALF) + rowoff, offB = (uint32_t)((2 + (wc >> 1)) * T2_HALF) + rowoff;
  bf16x8 fa[2][4], fb[2][4];                                       // [k-half of the step][tile]: one A half, all of B
  // fragment reads are inline asm (x2_common.h: lds_read_tr_frag_async): explicit lgkmcnt waits below
  auto readA = [&](uint32_t buf, int jh) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t a0 = buf + offA + ks * 8192 + (uint32_t)(((2 * (jh * 4 + j) + (fc4 >> 1)) ^ fsw) << 4);
        fa[ks][j] = lds_read_tr_frag_async(a0, a0 + 1024);
      }
  };
  auto readB = [&](uint32_t buf, int ih) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t a0 = buf + offB + ks * 8192 + (uint32_t)(((2 * ((wc & 1) * 4 + ih * 2 + i) + (fc4 >> 1)) ^ fsw) << 4);
        fb[ks][ih * 2 + i] = lds_read_tr_frag_async(a0, a0 + 1024);
      }
  };
  auto quad = [&](auto jh_, auto ih_) {                            // 16 MFMAs: A half jh (in fa) x B half ih
    constexpr int jh = decltype(jh_)::value, ih = decltype(ih_)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          // D[row = k][col = n]: lane holds n = fi and 4 consecutive k
          acc[ih * 2 + i][jh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[ks][ih * 2 + i], fa[ks][j], acc[ih * 2 + i][jh * 4 + j], 0, 0, 0);
  };
#define T2_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto ktile = [&](int kt, auto n1_, auto n2_) {                   // n1: step kt+1 exists, n2: step kt+2 exists
    constexpr bool n1 = decltype(n1_)::value, n2 = decltype(n2_)::value;
    const uint32_t buf = lds0 + (uint32_t)((kt & 1) * 4 * T2_HALF);
    // phase 1: A-lo (16 reads), B-lo (8), B-hi (8) requested; the first quadrant starts when the first 24 are back
    readA(buf, 0); readB(buf, 0); readB(buf, 1);
    if constexpr (n1) issue(kt + 1, S0{});
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); T2_FENCE();
    quad(S0{}, S0{}); T2_FENCE();
    // phase 2
    if constexpr (n1) issue(kt + 1, S1{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); T2_FENCE();
    quad(S0{}, S1{}); T2_FENCE();
    // phase 3: every wave holds its B fragments of this step (lgkmcnt(0) above) before B0 / B1 of this buffer are refilled
    if constexpr (n2) { asm volatile("s_barrier" ::: "memory"); issue(kt + 2, S2{}); }
    readA(buf, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); T2_FENCE();
    quad(S1{}, S1{}); T2_FENCE();
    // phase 4
    if constexpr (n2) issue(kt + 2, S3{});
    quad(S1{}, S0{}); T2_FENCE();
    if constexpr (n1) {
      // step kt+1 complete in LDS for this wave (only B0 / B1 of step kt+2 may still be in flight); then for all waves
      if constexpr (n2) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      T2_FENCE();
    }
  };
  issue(0, S0{}); issue(0, S1{}); issue(0, S2{}); issue(0, S3{});
  if (nk > 1) { issue(1, S2{}); issue(1, S3{}); asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  int kt = 0;
  for (; kt + 2 < nk; ++kt) ktile(kt, std::true_type{}, std::true_type{});
  if (kt + 1 < nk) { ktile(kt, std::true_type{}, std::false_type{}); ++kt; }
  ktile(kt, std::false_type{}, std::false_type{});

  if (gridDim.y > 1) {        // partial tile in fragment order: ws[split][tile][wave][i][j][lane] x float4
    float* dst = ws + ((size_t)blockIdx.y * total_tiles + tile) * 65536 + (size_t)wave * 8192 + lane * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<float4*>(dst + (i * 8 + j) * 256) = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = n0 + wr * 128 + j * 16 + fi;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + wc * 64 + i * 16 + fg * 4;
      if (k >= p.K) continue;
      float* dst = p.C + (size_t)n * p.ldc + k;
      float4 o = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (g.accumulate) { const float4 c = *reinterpret_cast<float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
      *reinterpret_cast<float4*>(dst) = o;
    }
  }
}
// C (+)= sum over splits of the partial tiles above.  One thread = one float4 of one tile (64 workgroups per tile).
__global__ __launch_bounds__(256) void gemm_tn256_reduce_kernel(GemmTNGroup g, const float* __restrict__ ws, int total_tiles, int split) {
  const int tile = blockIdx.x >> 6;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < g.count && tile >= g.p[i].tile_begin) pi = i;
  const TNProblem p = g.p[pi];
  int tnn, tkk;
  tile_coords(tile - p.tile_begin, (p.N + 255) / 256, p.tiles_k, 4, tnn, tkk);
  const int e = (blockIdx.x & 63) * 256 + threadIdx.x;             // float4 index inside the tile
  const int lane = e & 63, ij = (e >> 6) & 31, wave = e >> 11;
  const int i = ij >> 3, j = ij & 7, fi = lane & 15, fg = lane >> 4, wr = wave >> 2, wc = wave & 3;
  const int n = tnn * 256 + wr * 128 + j * 16 + fi, k = tkk * 256 + wc * 64 + i * 16 + fg * 4;
  const float* src = ws + (size_t)tile * 65536 + (size_t)e * 4;
  float4 o = *reinterpret_cast<const float4*>(src);
  for (int s_ = 1; s_ < split; ++s_) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)s_ * total_tiles * 65536);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
  }
  if (n >= p.N || k >= p.K) return;
  float* dst = p.C + (size_t)n * p.ldc + k;
  if (g.accumulate) { const float4 c = *reinterpret_cast<float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
  *reinterpret_cast<float4*>(dst) = o;
}

// problems: `count` rows of 11 int64: {A, B, C, Mc, N, K, lda, ldb, ldc, n_ld, k_ld}.
// split = 0: automatic.  ws / ws_floats: scratch for the split partial tiles of the 256x256 kernel (may be null: then
// it never splits).  The 256x256 kernel (any contraction length: a partial last step reads zeros) is used from 24 tiles
// of 256x256 up, the 128x128 kernel for small launches; there split > 1 adds atomically and needs accumulate = 1.
extern "C" int x2_gemm_tn_grouped(const int64_t* problems, int count, int accumulate, int split, float* ws, long ws_floats,
                                  void* stream) {
  X2_REQUIRE(count >= 1 && count <= 8, "x2_gemm_tn_grouped: count=%d not in [1,8]", count);
  X2_REQUIRE(split >= 0, "x2_gemm_tn_grouped: split=%d", split);
  GemmTNGroup g; g.count = count; g.accumulate = accumulate;
  int tiles = 0, tiles256 = 0, min_steps = 1 << 30;
  bool big = g_tune[5] != 1;
  for (int i = 0; i < count; ++i) {
    const int64_t* q = problems + i * 11;
    TNProblem& p = g.p[i];
    p.A = (const bf16_t*)q[0]; p.B = (const bf16_t*)q[1]; p.C = (float*)q[2];
    p.Mc = (int)q[3]; p.N = (int)q[4]; p.K = (int)q[5]; p.lda = (int)q[6]; p.ldb = (int)q[7]; p.ldc = (int)q[8];
    p.n_ld = (int)q[9]; p.k_ld = (int)q[10];
    X2_REQUIRE(p.Mc > 0 && p.N > 0 && p.K > 0, "x2_gemm_tn_grouped[%d]: empty problem", i);
    X2_REQUIRE(p.K % 4 == 0 && p.ldc % 4 == 0, "x2_gemm_tn_grouped[%d]: K, ldc must be multiples of 4", i);
    X2_REQUIRE(p.lda % 8 == 0 && p.ldb % 8 == 0 && p.n_ld % 8 == 0 && p.k_ld % 8 == 0 && p.n_ld >= 8 && p.k_ld >= 8,
               "x2_gemm_tn_grouped[%d]: rows must be 16-byte granular", i);
    X2_REQUIRE(p.n_ld <= p.lda && p.k_ld <= p.ldb, "x2_gemm_tn_grouped[%d]: n_ld/k_ld exceed leading dims", i);
    tiles += ((p.N + 127) / 128) * ((p.K + 127) / 128);
    tiles256 += ((p.N + 255) / 256) * ((p.K + 255) / 256);
    const int steps_i = (p.Mc + BK - 1) / BK;
    min_steps = steps_i < min_steps ? steps_i : min_steps;
  }
  // small launches (a head's single weight) stay on the 128x128 kernel: more, smaller workgroups
  // (an EXPLICIT split with a workspace keeps the 256x256 kernel and its fixed-order reduction however few tiles there are: the patch-embedding
  // weight gradient - 9 tiles, 196 steps - was the step's last atomic accumulation)
  if (g_tune[5] != 2 && tiles256 < 24 && !(split > 1 && ws)) big = false;
  if (big) {
    int t = 0;
    for (int i = 0; i < count; ++i) { g.p[i].tile_begin = t; g.p[i].tiles_k = (g.p[i].K + 255) / 256; t += ((g.p[i].N + 255) / 256) * g.p[i].tiles_k; }
    const int cus = x2_cus();
    int sp = split;
    if (sp == 0) {              // fullest last round of one-workgroup-per-CU slots, fewest partial tiles on a tie
      double best = -1.0; sp = 1;
      for (int c = 1; c <= 4 && min_steps / c >= 16; ++c) {     // a slice shorter than 16 steps costs more in partial tiles than it fills
        const long blocks = (long)t * c;
        const double fill = (double)blocks / (double)(((blocks + cus - 1) / cus) * cus);
        if (fill > best + 0.03) { best = fill; sp = c; }
      }
    }
    while (sp > 1 && (sp > min_steps || !ws || (long)sp * t * 65536 > ws_floats)) --sp;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS_BYTES);
      attr = true;
    }
    bool ragged = false;
    for (int i = 0; i < count; ++i) ragged = ragged || (g.p[i].Mc % BK != 0);
    if (ragged) hipLaunchKernelGGL(gemm_tn256_kernel<true>, dim3(t, sp), dim3(512), T2_LDS_BYTES, (hipStream_t)stream, g, ws, t);
    else hipLaunchKernelGGL(gemm_tn256_kernel<false>, dim3(t, sp), dim3(512), T2_LDS_BYTES, (hipStream_t)stream, g, ws, t);
    if (sp > 1) hipLaunchKernelGGL(gemm_tn256_reduce_kernel, dim3(t * 64), dim3(256), 0, (hipStream_t)stream, g, ws, t, sp);
    return x2_check_launch("x2_gemm_tn_grouped");
  }
  if (split == 0) split = 1;
  X2_REQUIRE(split == 1 || accumulate, "x2_gemm_tn_grouped: split>1 on the 128x128 kernel adds atomically: pass accumulate=1 and a defined C");
  int t = 0;
  for (int i = 0; i < count; ++i) { g.p[i].tile_begin = t; g.p[i].tiles_k = (g.p[i].K + 127) / 128; t += ((g.p[i].N + 127) / 128) * g.p[i].tiles_k; }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(t, split), dim3(256), GEMM_LDS_BYTES, (hipStream_t)stream, g);
  return x2_check_launch("x2_gemm_tn_grouped");
}
