// Fused multi-head attention (head dim 64) for gfx950, forward and backward, for the three
// attention flavours of the X^2-VLM step:
//   * BEiT-2 self-attention with a learned relative-position bias   (beit2.py:125-166)
//   * BERT self-attention with an additive padding mask             (xbert.py:322-415)
//   * BERT cross-attention text -> image tokens, image K/V shared by several text rows through
//     `kv_idx` (the reference recomputes K/V per pass: xbert.py:345-348; same values)
// Scores never touch HBM: per 16-query strip a wave keeps S^T = K.Q^T in MFMA accumulators
// (swapped operands, so every lane owns one query and softmax statistics are lane-local), runs
// the online softmax in registers, packs P straight into the next MFMA's operand, and reads V
// (and K^T, Q^T, dO^T in the backward) with the transposing LDS read ds_read_b64_tr_b16.
// Backward = two kernels (no atomics, deterministic): dQ (+ the dS stream the bias gradient is
// reduced from) and dK/dV - or ONE kernel where a (sequence, head) can be walked by one workgroup:
// attn_bwd_onepass_kernel (N <= 208), attn_bwd_onepass_grouped_kernel (text rows sharing an image's
// K/V), attn_bwd_onepass_long_kernel (208 < N <= 640: X2VLM-large); x2_attn_bwd picks.
#include "x2_common.h"
#include <cstdlib>
extern "C" int x2_tune_get(int key);

#define HD 64                 // head dim
#define KT 64                 // keys (or queries) per LDS tile
#define NEG_BIG (-1.0e30f)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#include <type_traits>
// v_exp_f32 as is: libm's exp2f wraps it in a denormal-range fix-up (compare, select, ldexp, multiply: five more VALU
// instructions per element in kernels whose inner loops are VALU-bound); arguments here are score - running max <= 0, and
// results below 2^-126 may flush to zero
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
using FullTile = std::true_type; using PartTile = std::false_type;

struct AttnArgs {
  const bf16_t *Q, *K, *V, *O, *dO;
  bf16_t *Out, *dQ, *dK, *dV, *dS;
  float *LSE, *Delta;
  long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;      // element strides: batch, row (token)
  long dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, do_bs, do_rs;
  int B, Bkv, H, Lq, Lk;
  float scale;
  const float* bias;  int bias_ld;     // [H][Lq][bias_ld]   (fwd, dQ)
  const float* biasT; int biasT_ld;    // [H][Lk][biasT_ld]  (dK/dV)
  const float* mask;  int mask_ld;     // [B][mask_ld] additive, per key
  const int* kv_idx;                   // [B] query batch -> kv batch (null: identity)
  const int* seq_off; const int* seq_ids;   // CSR: kv batch -> query batches using it (null: identity)
  int ds_ld;                           // dS: [B][H][Lq][ds_ld]
  DropSpec drop;                       // dropout on the attention probabilities (xbert.py:399), element index
                                       // ((b*H + h)*Lq + q) * round_up(Lk,64) + key
  int dbg;                             // ablation (probes/bench_attn.py): 1 no bias/mask loads, 2 no exp2, 4 no PV MFMAs, 8 no QK MFMAs;
                                       // 16 (not an ablation): bias / biasT hold bias x log2(e) (x2_relpos_bias with that scale)
  int head_dim;                        // the caller's head dimension: must be 64 (the only one these kernels are built for)
  const uint32_t* drop_epoch;          // device step counter mixed into drop.seed (x2_common.h drop_at_epoch), or NULL
  // filled by the entry points (callers pass zeros): logical grid (x = query / key tiles, y = heads, z = batches) and, when
  // grid_map > 0, the XCD-aware decode of a 1-D launch (attn_block below); grid_map = batch chunks per head
  int grid_nx, grid_ny, grid_nz, grid_map;
  int phase;                           // x2_attn_bwd: 0 = dQ (+ dS, Delta) then dK / dV; 1 = dQ (+ dS, Delta) only; 2 = dK / dV only (Delta as a
                                       // phase-1 call left it): lets a caller put the K/V-side gradients on another stream
  float* ws; long ws_floats;           // x2_attn_bwd: fp32 scratch for attn_bwd_onepass_long_kernel's dQ partials (B * H * ceil(Lq / 128) * 8192 floats),
                                       // or NULL / too small: the dQ + dK/dV pair runs instead
  float* colsum_ws;                    // x2_attn_bwd, forms 1 and 3 only (attn_bwd_onepass_kernel, attn_bwd_onepass_long_kernel), or NULL: [B][2][H * 64] fp32 partial column sums over a sequence's
                                       // rows of the STORED (bf16) dQ (k = 0) and dV (k = 1): the q / v bias gradient of a fused qkv projection is their sum
                                       // over B (x2_reduce_partials with nblk = B, nk = 2) - no pass over the [M, 3D] gradient
};

// Workgroup -> (tile, head, batch).  grid_map == 0: the 3-D grid as launched.  grid_map = C > 0 (the kernels with a relative-
// position bias): a 1-D grid decoded so that an XCD (hardware: workgroup id mod 8) works through whole (head, batch chunk)
// units, batch-major and tile-minor inside a unit.  Why: the bias of one head is [Lq][Lk] fp32 - 1.5 MB at N = 577, 23.6 MB
// for the 16 heads of X2VLM-large - and EVERY workgroup reads its [128][Lk] slice of it: 839 MB per launch at batch 32, more
// than Q, K, V and O together (455 MB), streamed from the Infinity Cache because with heads interleaved over the XCDs no
// 4 MB L2 keeps a head's bias between two images.  With a head pinned to an XCD its bias is fetched once and stays in that
// L2 for all images of the chunk, and the query tiles of one (image, head) follow each other, so K / V hit L2 as well.
__device__ __forceinline__ bool attn_block(const AttnArgs& a, int& bx, int& by, int& bz) {
  if (a.grid_map == 0) { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; return true; }
  const int w = blockIdx.x, xcd = w & 7, idx = w >> 3;
  const int C = a.grid_map, zc = (a.grid_nz + C - 1) / C, per_unit = zc * a.grid_nx;
  const int u = (idx / per_unit) * 8 + xcd, r = idx % per_unit;
  by = u / C; bz = (u % C) * zc + r / a.grid_nx; bx = r % a.grid_nx;
  return by < a.grid_ny && bz < a.grid_nz;
}

// Staging of a [64 rows][64 d] bf16 tile (rows clamped to `nrows-1`) HBM -> registers -> LDS in two halves, so the
// global loads of tile t+1 are in flight while tile t is multiplied (issue early / write late).  LDS image:
// chunk c (16 B) of row r at c ^ (r & 7).
template <int NT> struct TileRegs { u32x4 v[(512 + NT - 1) / NT]; };
template <int NT>
__device__ __forceinline__ void tile_load(TileRegs<NT>& t, const bf16_t* src, long rs, int row0, int nrows, int tid) {
#pragma unroll
  for (int i = 0; i < (512 + NT - 1) / NT; ++i) {
    const int c = tid + i * NT;
    if (c < 512) {
      int gr = row0 + (c >> 3); gr = gr < nrows ? gr : nrows - 1;
      t.v[i] = *reinterpret_cast<const u32x4*>(src + (long)gr * rs + (c & 7) * 8);
    }
  }
}
template <int NT>
__device__ __forceinline__ void tile_store(const TileRegs<NT>& t, char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < (512 + NT - 1) / NT; ++i) {
    const int c = tid + i * NT;
    if (c < 512) { const int r = c >> 3, ch = c & 7; *reinterpret_cast<u32x4*>(lds + r * 128 + ((ch ^ (r & 7)) << 4)) = t.v[i]; }
  }
}

// A-operand fragment, rows = tile rows (16 per MFMA tile), contraction = d
__device__ __forceinline__ bf16x8 frag_rows(uint32_t tile, int row, int chunk) {
  return lds_read_b128(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
}
// A-operand fragment of the TRANSPOSED tile: rows = d (16 per MFMA tile `dt`), contraction = tile rows
// in the slot order (g, j<4) -> row 32s + 4g + j, (g, j>=4) -> row 32s + 16 + 4g + (j-4)
__device__ __forceinline__ bf16x8 frag_cols(uint32_t tile, int s, int dt, int lane) {
  const int fi = lane & 15, g = lane >> 4, r = fi >> 2, c4 = fi & 3;
  const int row = 32 * s + 4 * g + r;
  const uint32_t a0 = tile + row * 128 + (((2 * dt + (c4 >> 1)) ^ (row & 7)) << 4) + (c4 & 1) * 8;
  // row + 16 has the same (row & 7)
  return lds_read_tr_frag(a0, a0 + 16 * 128);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  u32x4 u{pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])};
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ float group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// scores (log2 domain) of one S^T tile row-group for this lane: keys key0..key0+3, query q
__device__ __forceinline__ f32x4 add_bias_mask(f32x4 s, const AttnArgs& a, int h, int b, int q, int key0, float sc2) {
  float4 bb{0.f, 0.f, 0.f, 0.f}, mm{0.f, 0.f, 0.f, 0.f};
  if (a.bias && !(a.dbg & 1)) bb = *reinterpret_cast<const float4*>(a.bias + ((long)h * a.Lq + q) * a.bias_ld + key0);
  if (a.mask && !(a.dbg & 1)) mm = *reinterpret_cast<const float4*>(a.mask + (long)b * a.mask_ld + key0);
  f32x4 o;
  o[0] = key0 + 0 < a.Lk ? s[0] * sc2 + (bb.x + mm.x) * LOG2E : NEG_BIG;
  o[1] = key0 + 1 < a.Lk ? s[1] * sc2 + (bb.y + mm.y) * LOG2E : NEG_BIG;
  o[2] = key0 + 2 < a.Lk ? s[2] * sc2 + (bb.z + mm.z) * LOG2E : NEG_BIG;
  o[3] = key0 + 3 < a.Lk ? s[3] * sc2 + (bb.w + mm.w) * LOG2E : NEG_BIG;
  return o;
}

// the same with the bias / mask values already in registers (loaded at the top of the key tile, so that their L2
// latency is covered by the QK^T MFMAs instead of sitting between them and the softmax)
// BL2: the bias is already in log2 units (x log2 e, x2_relpos_bias with scale = log2 e) and there is no mask: one fma per
// score instead of add + two multiplies + add (these loops are VALU-bound: ISA of the forward kernel)
template <bool FULL = false, bool BL2 = false>
__device__ __forceinline__ f32x4 apply_bias_mask(f32x4 s, float4 bb, float4 mm, int key0, int Lk, float sc2) {
  f32x4 o;
  if constexpr (BL2) {
    o[0] = fmaf(s[0], sc2, bb.x); o[1] = fmaf(s[1], sc2, bb.y); o[2] = fmaf(s[2], sc2, bb.z); o[3] = fmaf(s[3], sc2, bb.w);
    if constexpr (!FULL) {
      o[0] = key0 + 0 < Lk ? o[0] : NEG_BIG; o[1] = key0 + 1 < Lk ? o[1] : NEG_BIG;
      o[2] = key0 + 2 < Lk ? o[2] : NEG_BIG; o[3] = key0 + 3 < Lk ? o[3] : NEG_BIG;
    }
    return o;
  }
  if constexpr (FULL) {       // every key of the tile exists: no index select
    o[0] = s[0] * sc2 + (bb.x + mm.x) * LOG2E; o[1] = s[1] * sc2 + (bb.y + mm.y) * LOG2E;
    o[2] = s[2] * sc2 + (bb.z + mm.z) * LOG2E; o[3] = s[3] * sc2 + (bb.w + mm.w) * LOG2E;
    return o;
  }
  o[0] = key0 + 0 < Lk ? s[0] * sc2 + (bb.x + mm.x) * LOG2E : NEG_BIG;
  o[1] = key0 + 1 < Lk ? s[1] * sc2 + (bb.y + mm.y) * LOG2E : NEG_BIG;
  o[2] = key0 + 2 < Lk ? s[2] * sc2 + (bb.z + mm.z) * LOG2E : NEG_BIG;
  o[3] = key0 + 3 < Lk ? s[3] * sc2 + (bb.w + mm.w) * LOG2E : NEG_BIG;
  return o;
}
template <int QG, bool BL2 = false>
__device__ __forceinline__ void load_bias_mask(const AttnArgs& a, int h, int b, const int (&q)[QG], int key_base, int g, int nsub,
                                               float4 (&bb)[QG][4], float4 (&mm)[4]) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    mm[nt] = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gq = 0; gq < QG; ++gq) bb[gq][nt] = float4{0.f, 0.f, 0.f, 0.f};
  }
  if (a.bias && !(a.dbg & 1)) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int gq = 0; gq < QG; ++gq)
        if (nt < nsub) bb[gq][nt] = *reinterpret_cast<const float4*>(a.bias + ((long)h * a.Lq + q[gq]) * a.bias_ld + key_base + nt * 16 + g * 4);
    // a bias in log2 units (dbg bit 4) reaching a kernel that has no one-fma form (the callers of this function multiply by
    // log2 e themselves): back to natural units here, a wave-uniform branch outside the per-score arithmetic
    if (!BL2 && (a.dbg & 16)) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) { bb[gq][nt].x *= LN2; bb[gq][nt].y *= LN2; bb[gq][nt].z *= LN2; bb[gq][nt].w *= LN2; }
    }
  }
  if (a.mask && !(a.dbg & 1)) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (nt < nsub) mm[nt] = *reinterpret_cast<const float4*>(a.mask + (long)b * a.mask_ld + key_base + nt * 16 + g * 4);
  }
}

// ------------------------------------------------------------------------------------------ forward
// QW waves per workgroup, QG groups of 16 queries per wave (K / V^T fragments read from LDS once serve QG MFMAs).
// RES: all K / V tiles of the (batch, head) are resident in LDS (Lk <= 256): one load phase and one barrier per
// workgroup instead of one per key tile - these kernels are latency-bound, not MFMA-bound, at N = 197 / 30.
// NS: LDS slots.  Resident form: one per key tile (4 covers Lk <= 256; 1 for Lk <= 64, the 30-token text sequences: 16 KB
// per workgroup instead of 32-64 KB lets 10 of the 2-wave workgroups share a CU instead of 5, and these launches are a
// serial load -> multiply -> store chain per workgroup whose only latency hiding is other workgroups).
template <int QW, int QG, bool RES, int NS = (RES ? 4 : 2), int WPS = (QG > 1 ? 2 : 4), bool BL2 = false>     // WPS: waves per SIMD the registers must allow
__global__ __launch_bounds__(64 * QW, WPS) void attn_fwd_kernel(AttnArgs a) {
  const DropSpec drop_ = drop_at_epoch(a.drop, a.drop_epoch);
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) char smem[NS][2 * KT * 128];   // {K tile, V tile} per slot
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  int bx_, h, b;
  if (!attn_block(a, bx_, h, b)) return;
  const int bk = a.kv_idx ? a.kv_idx[b] : b;
  const bf16_t* Kp = a.K + bk * a.k_bs + h * HD;
  const bf16_t* Vp = a.V + bk * a.v_bs + h * HD;
  const float sc2 = a.scale * LOG2E;
  const int lkp = (a.Lk + 63) & ~63;

  int q[QG]; bool qok[QG];
  bf16x8 qf[QG][2];
  f32x4 o[QG][4];
  float m_i[QG], l_i[QG];
#pragma unroll
  for (int gq = 0; gq < QG; ++gq) {
    const int q0 = (bx_ * QW * QG + wave * QG + gq) * 16;
    qok[gq] = q0 + fi < a.Lq;
    q[gq] = min(q0 + fi, a.Lq - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[gq][ks] = *reinterpret_cast<const bf16x8*>(a.Q + b * a.q_bs + (long)q[gq] * a.q_rs + h * HD + ks * 32 + g * 8);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[gq][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_i[gq] = NEG_BIG; l_i[gq] = 0.f;
  }

  const int nkt = (a.Lk + KT - 1) / KT;
  TileRegs<NT> rk, rv;
  if (RES) {
    // all (<= 4) key tiles are requested before the first one is written to LDS: one exposed HBM/L2 latency per
    // workgroup instead of one per tile
    TileRegs<NT> rka[NS], rva[NS];
#pragma unroll
    for (int kt = 0; kt < NS; ++kt)
      if (kt < nkt) { tile_load<NT>(rka[kt], Kp, a.k_rs, kt * KT, a.Lk, tid); tile_load<NT>(rva[kt], Vp, a.v_rs, kt * KT, a.Lk, tid); }
#pragma unroll
    for (int kt = 0; kt < NS; ++kt)
      if (kt < nkt) { tile_store<NT>(rka[kt], smem[kt], tid); tile_store<NT>(rva[kt], smem[kt] + KT * 128, tid); }
  } else {
    tile_load<NT>(rk, Kp, a.k_rs, 0, a.Lk, tid);
    tile_load<NT>(rv, Vp, a.v_rs, 0, a.Lk, tid);
    tile_store<NT>(rk, smem[0], tid);
    tile_store<NT>(rv, smem[0] + KT * 128, tid);
  }
  __syncthreads();
  // N = 197 is 12.3 sixteen-row MFMA tiles, not 16: a wave whose 16*QG queries all lie past Lq has nothing to do (no
  // barrier follows in the resident form), and 16-key sub-tiles past Lk are skipped instead of multiplied and masked
  const bool idle = (bx_ * QW * QG + wave * QG) * 16 >= a.Lq;
  if (RES && idle) return;
  // One key tile.  FULL (compile time): all 64 keys exist and this wave has queries - the 16-key sub-tile tests and the
  // key-index selects fold away and the tile is straight-line code (the tests used to put every MFMA in a basic block of its
  // own: nothing could be scheduled across them, and these loops are VALU-bound); only the last tile of a sequence, and
  // waves without queries, take the general form.
  auto tile = [&](int kt, auto full_) {
    constexpr bool FULL = decltype(full_)::value;
    const uint32_t ktile = lds_addr(smem[RES ? kt : (kt & 1)]), vtile = ktile + KT * 128;
    if (!RES && kt + 1 < nkt) {             // next tile's HBM loads fly under this tile's MFMAs
      tile_load<NT>(rk, Kp, a.k_rs, (kt + 1) * KT, a.Lk, tid);
      tile_load<NT>(rv, Vp, a.v_rs, (kt + 1) * KT, a.Lk, tid);
    }
    const int nsub = FULL ? 4 : (idle ? 0 : min(4, (a.Lk - kt * KT + 15) >> 4));      // valid 16-key sub-tiles of this tile (wave-uniform)
    float4 bbv[QG][4], mmv[4];
    load_bias_mask<QG, BL2>(a, h, b, q, kt * KT, g, nsub, bbv, mmv);
    f32x4 st[QG][4];
#pragma unroll
    for (int gq = 0; gq < QG; ++gq)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) st[gq][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if ((a.dbg & 8) || nt >= nsub) continue;
        const bf16x8 kfr = frag_rows(ktile, nt * 16 + fi, ks * 4 + g);
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) st[gq][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[gq][ks], st[gq][nt], 0, 0, 0);
      }
    bf16x8 pf[QG][2];
#pragma unroll
    for (int gq = 0; gq < QG; ++gq) {
      float mx = NEG_BIG;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt >= nsub) continue;
        st[gq][nt] = apply_bias_mask<FULL, BL2>(st[gq][nt], bbv[gq][nt], mmv[nt], kt * KT + nt * 16 + g * 4, a.Lk, sc2);
        mx = fmaxf(fmaxf(mx, fmaxf(st[gq][nt][0], st[gq][nt][1])), fmaxf(st[gq][nt][2], st[gq][nt][3]));
      }
      mx = group_max(mx);
      const float m_new = fmaxf(m_i[gq], mx);
      const float alpha = fast_exp2(m_i[gq] - m_new);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt >= nsub) continue;           // skipped sub-tiles keep P = 0 (st was zero-initialised)
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[gq][nt][r] = (a.dbg & 2) ? (st[gq][nt][r] - m_new) * 0.001f : fast_exp2(st[gq][nt][r] - m_new); rs += st[gq][nt][r]; }
      }
      l_i[gq] = l_i[gq] * alpha + group_sum(rs);
      m_i[gq] = m_new;
      if (drop_.thr16) {      // normalisation uses the undropped sum; only the P that multiplies V is dropped
        const uint32_t e0 = (uint32_t)(((long)b * a.H + h) * a.Lq + q[gq]) * (uint32_t)lkp + (uint32_t)(kt * KT + g * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          if (nt >= nsub) continue;
          float dm[4];
          drop_mul4(drop_, e0 + nt * 16, dm);
          st[gq][nt][0] *= dm[0]; st[gq][nt][1] *= dm[1]; st[gq][nt][2] *= dm[2]; st[gq][nt][3] *= dm[3];
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[gq][dt] *= alpha;
      pf[gq][0] = pack8(st[gq][0], st[gq][1]);
      pf[gq][1] = pack8(st[gq][2], st[gq][3]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if ((a.dbg & 4) || 2 * s2 >= nsub) continue;
        const bf16x8 vfr = frag_cols(vtile, s2, dt, lane);
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) o[gq][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, pf[gq][s2], o[gq][dt], 0, 0, 0);
      }
    if (!RES) {
      if (kt + 1 < nkt) {
        tile_store<NT>(rk, smem[(kt + 1) & 1], tid);
        tile_store<NT>(rv, smem[(kt + 1) & 1] + KT * 128, tid);
      }
      __syncthreads();
    }
  };
  for (int kt = 0; kt < nkt; ++kt) {
    if (!idle && (kt + 1) * KT <= a.Lk) tile(kt, FullTile{});
    else tile(kt, PartTile{});
  }
#pragma unroll
  for (int gq = 0; gq < QG; ++gq) {
    if (!qok[gq]) continue;
    const float inv = 1.0f / l_i[gq];
    bf16_t* op = a.Out + b * a.o_bs + (long)q[gq] * a.o_rs + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<u32x2*>(op + dt * 16) = u32x2{pack_bf16(o[gq][dt][0] * inv, o[gq][dt][1] * inv), pack_bf16(o[gq][dt][2] * inv, o[gq][dt][3] * inv)};
    if (g == 0) a.LSE[((long)b * a.H + h) * a.Lq + q[gq]] = m_i[gq] + log2f(l_i[gq]);   // log2 domain
  }
}

// ------------------------------------------------------------------------------------------ backward: dQ (+ dS)
template <int QW, int QG, bool RES, int NS = (RES ? 4 : 2), int WPS = (QG > 1 ? 2 : 4), bool BL2 = false>
__global__ __launch_bounds__(64 * QW, WPS) void attn_bwd_dq_kernel(AttnArgs a) {
  const DropSpec drop_ = drop_at_epoch(a.drop, a.drop_epoch);
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) char smem[NS][2 * KT * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  int bx_, h, b;
  if (!attn_block(a, bx_, h, b)) return;
  const int bk = a.kv_idx ? a.kv_idx[b] : b;
  const bf16_t* Kp = a.K + bk * a.k_bs + h * HD;
  const bf16_t* Vp = a.V + bk * a.v_bs + h * HD;
  const float sc2 = a.scale * LOG2E;
  const int lkp = (a.Lk + 63) & ~63;

  int q[QG]; bool qok[QG];
  bf16x8 qf[QG][2], dof[QG][2];
  float delta[QG], lse[QG];
  f32x4 dq[QG][4];
#pragma unroll
  for (int gq = 0; gq < QG; ++gq) {
    const int q0 = (bx_ * QW * QG + wave * QG + gq) * 16;
    qok[gq] = q0 + fi < a.Lq;
    q[gq] = min(q0 + fi, a.Lq - 1);
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[gq][ks] = *reinterpret_cast<const bf16x8*>(a.Q + b * a.q_bs + (long)q[gq] * a.q_rs + h * HD + ks * 32 + g * 8);
      dof[gq][ks] = *reinterpret_cast<const bf16x8*>(a.dO + b * a.do_bs + (long)q[gq] * a.do_rs + h * HD + ks * 32 + g * 8);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.O + b * a.o_bs + (long)q[gq] * a.o_rs + h * HD + ks * 32 + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += bf2f((bf16_t)dof[gq][ks][e]) * bf2f((bf16_t)of[e]);
    }
    delta[gq] = group_sum(dl);
    lse[gq] = a.LSE[((long)b * a.H + h) * a.Lq + q[gq]];
    if (qok[gq] && g == 0) a.Delta[((long)b * a.H + h) * a.Lq + q[gq]] = delta[gq];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[gq][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int nkt = (a.Lk + KT - 1) / KT;
  TileRegs<NT> rk, rv;
  if (RES) {
    // all (<= 4) key tiles are requested before the first one is written to LDS: one exposed HBM/L2 latency per
    // workgroup instead of one per tile
    TileRegs<NT> rka[NS], rva[NS];
#pragma unroll
    for (int kt = 0; kt < NS; ++kt)
      if (kt < nkt) { tile_load<NT>(rka[kt], Kp, a.k_rs, kt * KT, a.Lk, tid); tile_load<NT>(rva[kt], Vp, a.v_rs, kt * KT, a.Lk, tid); }
#pragma unroll
    for (int kt = 0; kt < NS; ++kt)
      if (kt < nkt) { tile_store<NT>(rka[kt], smem[kt], tid); tile_store<NT>(rva[kt], smem[kt] + KT * 128, tid); }
  } else {
    tile_load<NT>(rk, Kp, a.k_rs, 0, a.Lk, tid);
    tile_load<NT>(rv, Vp, a.v_rs, 0, a.Lk, tid);
    tile_store<NT>(rk, smem[0], tid);
    tile_store<NT>(rv, smem[0] + KT * 128, tid);
  }
  __syncthreads();
  const bool idle = (bx_ * QW * QG + wave * QG) * 16 >= a.Lq;       // see attn_fwd_kernel
  if (RES && idle) return;
  auto tile = [&](int kt, auto full_) {          // FULL: see attn_fwd_kernel
    constexpr bool FULL = decltype(full_)::value;
    const uint32_t ktile = lds_addr(smem[RES ? kt : (kt & 1)]), vtile = ktile + KT * 128;
    if (!RES && kt + 1 < nkt) {
      tile_load<NT>(rk, Kp, a.k_rs, (kt + 1) * KT, a.Lk, tid);
      tile_load<NT>(rv, Vp, a.v_rs, (kt + 1) * KT, a.Lk, tid);
    }
    const int nsub = FULL ? 4 : (idle ? 0 : min(4, (a.Lk - kt * KT + 15) >> 4));
    float4 bbv[QG][4], mmv[4];
    load_bias_mask<QG, BL2>(a, h, b, q, kt * KT, g, nsub, bbv, mmv);
    f32x4 ds[QG][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      if (nt >= nsub) {
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) ds[gq][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        continue;
      }
      f32x4 s[QG], dp[QG];
#pragma unroll
      for (int gq = 0; gq < QG; ++gq) { s[gq] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[gq] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kfr = frag_rows(ktile, nt * 16 + fi, ks * 4 + g), vfr = frag_rows(vtile, nt * 16 + fi, ks * 4 + g);
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) {
          s[gq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[gq][ks], s[gq], 0, 0, 0);
          dp[gq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, dof[gq][ks], dp[gq], 0, 0, 0);
        }
      }
      const int key0 = kt * KT + nt * 16 + g * 4;
#pragma unroll
      for (int gq = 0; gq < QG; ++gq) {
        s[gq] = apply_bias_mask<FULL, BL2>(s[gq], bbv[gq][nt], mmv[nt], key0, a.Lk, sc2);
        if (drop_.thr16) {
          float dm[4];
          drop_mul4(drop_, (uint32_t)(((long)b * a.H + h) * a.Lq + q[gq]) * (uint32_t)lkp + (uint32_t)key0, dm);
          dp[gq][0] *= dm[0]; dp[gq][1] *= dm[1]; dp[gq][2] *= dm[2]; dp[gq][3] *= dm[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[gq][nt][r] = fast_exp2(s[gq][r] - lse[gq]) * (dp[gq][r] - delta[gq]);
        if (a.dS && qok[gq] && key0 < a.ds_ld)
          *reinterpret_cast<u32x2*>(a.dS + (((long)b * a.H + h) * a.Lq + q[gq]) * a.ds_ld + key0) =
              u32x2{pack_bf16(ds[gq][nt][0], ds[gq][nt][1]), pack_bf16(ds[gq][nt][2], ds[gq][nt][3])};
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 dsf[QG];
#pragma unroll
      for (int gq = 0; gq < QG; ++gq) dsf[gq] = pack8(ds[gq][2 * s2], ds[gq][2 * s2 + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        if (2 * s2 >= nsub) continue;
        const bf16x8 ktr = frag_cols(ktile, s2, dt, lane);
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) dq[gq][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr, dsf[gq], dq[gq][dt], 0, 0, 0);
      }
    }
    if (!RES) {
      if (kt + 1 < nkt) {
        tile_store<NT>(rk, smem[(kt + 1) & 1], tid);
        tile_store<NT>(rv, smem[(kt + 1) & 1] + KT * 128, tid);
      }
      __syncthreads();
    }
  };
  // (the straight-line FULL form of the forward kernels costs registers here: under this kernel's 128-VGPR bound it spills, and
  // at 256 VGPRs the lost occupancy outweighs it - measured 639 -> 1492 / 667 us for dQ + dK/dV at N = 577; general form only)
  for (int kt = 0; kt < nkt; ++kt) tile(kt, PartTile{});
#pragma unroll
  for (int gq = 0; gq < QG; ++gq) {
    if (!qok[gq]) continue;
    bf16_t* op = a.dQ + b * a.dq_bs + (long)q[gq] * a.dq_rs + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<u32x2*>(op + dt * 16) = u32x2{pack_bf16(dq[gq][dt][0] * a.scale, dq[gq][dt][1] * a.scale),
                                                       pack_bf16(dq[gq][dt][2] * a.scale, dq[gq][dt][3] * a.scale)};
  }
}

// ------------------------------------------------------------------------------------------ grouped form (shared K/V)
// Cross-attention of the fusion stack: several text rows attend to the same image (the 4-pass batch: ~4 rows per image).
// One workgroup per (image, head) keeps that image's K / V resident in LDS and walks the concatenated queries of all rows
// that use it (CSR seq_off / seq_ids), 16 per wave and pass - the image's K / V leave L2 once per head instead of once
// per text row, and the 30-query rows fill 8-wave workgroups.  The batch index is per lane (rows of different sequences
// share a wave).  Backward (dQ) only: the forward twin of this kernel was 62 vs 68 us in isolation but slower in the step -
// round 2: -3 % with three streams; round 4, tail segment alone on the GPU: 23.63-23.70 vs 23.42 ms per base step
// (profiles/r05i_knob_ab.txt) - and was removed.

template <int QW>
__global__ __launch_bounds__(64 * QW, 32 / QW) void attn_bwd_dq_grouped_kernel(AttnArgs a) {
  const DropSpec drop_ = drop_at_epoch(a.drop, a.drop_epoch);
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) char smem[4][2 * KT * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, bk = blockIdx.z;
  const int sb = a.seq_off[bk], nrows = (a.seq_off[bk + 1] - sb) * a.Lq;
  if (nrows == 0) return;
  const bf16_t* Kp = a.K + bk * a.k_bs + h * HD;
  const bf16_t* Vp = a.V + bk * a.v_bs + h * HD;
  const float sc2 = a.scale * LOG2E;
  const int lkp = (a.Lk + 63) & ~63;
  const int nkt = (a.Lk + KT - 1) / KT;
  {
    TileRegs<NT> rka[4], rva[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (kt < nkt) { tile_load<NT>(rka[kt], Kp, a.k_rs, kt * KT, a.Lk, tid); tile_load<NT>(rva[kt], Vp, a.v_rs, kt * KT, a.Lk, tid); }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
      if (kt < nkt) { tile_store<NT>(rka[kt], smem[kt], tid); tile_store<NT>(rva[kt], smem[kt] + KT * 128, tid); }
  }
  __syncthreads();
  for (int base = wave * 16; base < nrows; base += QW * 16) {
    const bool qok = base + fi < nrows;
    const int v = min(base + fi, nrows - 1), sq = v / a.Lq, q = v - sq * a.Lq;
    const int b = a.seq_ids[sb + sq];
    bf16x8 qf[2], dof[2];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(a.Q + b * a.q_bs + (long)q * a.q_rs + h * HD + ks * 32 + g * 8);
      dof[ks] = *reinterpret_cast<const bf16x8*>(a.dO + b * a.do_bs + (long)q * a.do_rs + h * HD + ks * 32 + g * 8);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.O + b * a.o_bs + (long)q * a.o_rs + h * HD + ks * 32 + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += bf2f((bf16_t)dof[ks][e]) * bf2f((bf16_t)of[e]);
    }
    const float delta = group_sum(dl);
    const float lse = a.LSE[((long)b * a.H + h) * a.Lq + q];
    if (qok && g == 0) a.Delta[((long)b * a.H + h) * a.Lq + q] = delta;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; ++kt) {
      const uint32_t ktile = lds_addr(smem[kt]), vtile = ktile + KT * 128;
      const int nsub = min(4, (a.Lk - kt * KT + 15) >> 4);
      float4 mm[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        mm[nt] = (a.mask && nt < nsub) ? *reinterpret_cast<const float4*>(a.mask + (long)b * a.mask_ld + kt * KT + nt * 16 + g * 4)
                                       : float4{0.f, 0.f, 0.f, 0.f};
      f32x4 ds[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt >= nsub) { ds[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(ktile, nt * 16 + fi, ks * 4 + g), qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(vtile, nt * 16 + fi, ks * 4 + g), dof[ks], dp, 0, 0, 0);
        }
        const int key0 = kt * KT + nt * 16 + g * 4;
        s = apply_bias_mask(s, float4{0.f, 0.f, 0.f, 0.f}, mm[nt], key0, a.Lk, sc2);
        if (drop_.thr16) {
          float dm[4];
          drop_mul4(drop_, (uint32_t)(((long)b * a.H + h) * a.Lq + q) * (uint32_t)lkp + (uint32_t)key0, dm);
          dp[0] *= dm[0]; dp[1] *= dm[1]; dp[2] *= dm[2]; dp[3] *= dm[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[nt][r] = fast_exp2(s[r] - lse) * (dp[r] - delta);
      }
      const bf16x8 dsf[2] = {pack8(ds[0], ds[1]), pack8(ds[2], ds[3])};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          if (2 * s2 >= nsub) continue;
          dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(ktile, s2, dt, lane), dsf[s2], dq[dt], 0, 0, 0);
        }
    }
    if (qok) {
      bf16_t* op = a.dQ + b * a.dq_bs + (long)q * a.dq_rs + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<u32x2*>(op + dt * 16) = u32x2{pack_bf16(dq[dt][0] * a.scale, dq[dt][1] * a.scale),
                                                         pack_bf16(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
    }
  }
}

// ------------------------------------------------------------------------------------------ strip-walking resident forms
// Selected by default for Lk <= WALK_ROWS without K/V sharing (X2_ATTN_VARIANT bits 4096 forward / 8192 dQ; 0 = the two-workgroup
// resident kernels above).
// The resident kernels above launch ceil(Lq / 128) workgroups per (sequence, head) and each loads ALL of K and V: at
// N = 197 that is two workgroups, K / V read twice (126 MB moved per forward launch for 77 MB of operands) and three rounds
// of 64 KB workgroups.  Here ONE four-wave workgroup per (sequence, head) keeps K / V as sixteen-row strips - WALK_ROWS =
// 208 rows each, 52 KB, so three workgroups share a CU and all 768 (image, head) pairs of the base step are resident at
// once - and every wave walks the 16-query strips wave, wave + 4, ... in sequence (the loop of the grouped kernels, plus
// the relative-position bias and the dS stream).  LDS image: the operand that is read TRANSPOSED (V in the forward, K in
// dQ) comes first and the other one behind it, so that the transposing fragment reads of a partial last key tile (rows up
// to 16 past the 208) land on real rows of the other operand: finite values that meet P = 0 / dS = 0.
#define WALK_ROWS 208
#define WALK_BYTES (WALK_ROWS * 128)
template <int NT>
__device__ __forceinline__ void tile_store_bounded(const TileRegs<NT>& t, char* region, int row0, int tid) {
#pragma unroll
  for (int i = 0; i < (512 + NT - 1) / NT; ++i) {
    const int c = tid + i * NT;
    if (c < 512) {
      const int r = row0 + (c >> 3), ch = c & 7;
      if (r < WALK_ROWS) *reinterpret_cast<u32x4*>(region + r * 128 + ((ch ^ (r & 7)) << 4)) = t.v[i];
    }
  }
}
template <int NT>
__device__ __forceinline__ void walk_load_kv(const AttnArgs& a, const bf16_t* Kp, const bf16_t* Vp, char* kreg, char* vreg, int nkt, int tid) {
  TileRegs<NT> rka[4], rva[4];
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
    if (kt < nkt) { tile_load<NT>(rka[kt], Kp, a.k_rs, kt * KT, a.Lk, tid); tile_load<NT>(rva[kt], Vp, a.v_rs, kt * KT, a.Lk, tid); }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
    if (kt < nkt) { tile_store_bounded<NT>(rva[kt], vreg, kt * KT, tid); tile_store_bounded<NT>(rka[kt], kreg, kt * KT, tid); }
}

template <int QW, bool BL2 = false>
__global__ __launch_bounds__(64 * QW, 3) void attn_fwd_walk_kernel(AttnArgs a) {
  const DropSpec drop_ = drop_at_epoch(a.drop, a.drop_epoch);
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) char smem[2 * WALK_BYTES];        // V rows | K rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  int bx_, h, b;
  if (!attn_block(a, bx_, h, b)) return;
  const float sc2 = a.scale * LOG2E;
  const int lkp = (a.Lk + 63) & ~63;
  const int nkt = (a.Lk + KT - 1) / KT;
  walk_load_kv<NT>(a, a.K + b * a.k_bs + h * HD, a.V + b * a.v_bs + h * HD, smem + WALK_BYTES, smem, nkt, tid);
  __syncthreads();
  const uint32_t vbase = lds_addr(smem), kbase = vbase + WALK_BYTES;      // transposed reads (V) may run 16 rows over: into K rows
  for (int base = wave * 16; base < a.Lq; base += QW * 16) {              // no barrier below: waves run on their own
    const bool qok = base + fi < a.Lq;
    const int q1[1] = {min(base + fi, a.Lq - 1)};
    const int q = q1[0];
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[ks] = *reinterpret_cast<const bf16x8*>(a.Q + b * a.q_bs + (long)q * a.q_rs + h * HD + ks * 32 + g * 8);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_i = NEG_BIG, l_i = 0.f;
    auto tile = [&](int kt, auto full_) {          // FULL: see attn_fwd_kernel
      constexpr bool FULL = decltype(full_)::value;
      const uint32_t ktile = kbase + kt * KT * 128, vtile = vbase + kt * KT * 128;
      const int nsub = FULL ? 4 : min(4, (a.Lk - kt * KT + 15) >> 4);
      float4 bbv[1][4], mmv[4];
      load_bias_mask<1, BL2>(a, h, b, q1, kt * KT, g, nsub, bbv, mmv);
      f32x4 st[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) st[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (nt >= nsub) continue;
          st[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(ktile, nt * 16 + fi, ks * 4 + g), qf[ks], st[nt], 0, 0, 0);
        }
      float mx = NEG_BIG;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt >= nsub) continue;
        st[nt] = apply_bias_mask<FULL, BL2>(st[nt], bbv[0][nt], mmv[nt], kt * KT + nt * 16 + g * 4, a.Lk, sc2);
        mx = fmaxf(fmaxf(mx, fmaxf(st[nt][0], st[nt][1])), fmaxf(st[nt][2], st[nt][3]));
      }
      mx = group_max(mx);
      const float m_new = fmaxf(m_i, mx), alpha = fast_exp2(m_i - m_new);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt >= nsub) continue;               // skipped sub-tiles keep P = 0
#pragma unroll
        for (int r = 0; r < 4; ++r) { st[nt][r] = fast_exp2(st[nt][r] - m_new); rs += st[nt][r]; }
      }
      l_i = l_i * alpha + group_sum(rs);
      m_i = m_new;
      if (drop_.thr16) {
        const uint32_t e0 = (uint32_t)(((long)b * a.H + h) * a.Lq + q) * (uint32_t)lkp + (uint32_t)(kt * KT + g * 4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          if (nt >= nsub) continue;
          float dm[4];
          drop_mul4(drop_, e0 + nt * 16, dm);
          st[nt][0] *= dm[0]; st[nt][1] *= dm[1]; st[nt][2] *= dm[2]; st[nt][3] *= dm[3];
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
      const bf16x8 pf[2] = {pack8(st[0], st[1]), pack8(st[2], st[3])};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          if (2 * s2 >= nsub) continue;
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(vtile, s2, dt, lane), pf[s2], o[dt], 0, 0, 0);
        }
    };
    for (int kt = 0; kt < nkt; ++kt) {
      if ((kt + 1) * KT <= a.Lk) tile(kt, FullTile{});
      else tile(kt, PartTile{});
    }
    if (qok) {
      const float inv = 1.0f / l_i;
      bf16_t* op = a.Out + b * a.o_bs + (long)q * a.o_rs + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<u32x2*>(op + dt * 16) = u32x2{pack_bf16(o[dt][0] * inv, o[dt][1] * inv), pack_bf16(o[dt][2] * inv, o[dt][3] * inv)};
      if (g == 0) a.LSE[((long)b * a.H + h) * a.Lq + q] = m_i + log2f(l_i);   // log2 domain
    }
  }
}

template <int QW, bool BL2 = false>
__global__ __launch_bounds__(64 * QW, 3) void attn_bwd_dq_walk_kernel(AttnArgs a) {
  const DropSpec drop_ = drop_at_epoch(a.drop, a.drop_epoch);
  constexpr int NT = 64 * QW;
  __shared__ __attribute__((aligned(16))) char smem[2 * WALK_BYTES];        // K rows | V rows (here the transposed reads are K's)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  int bx_, h, b;
  if (!attn_block(a, bx_, h, b)) return;
  const float sc2 = a.scale * LOG2E;
  const int lkp = (a.Lk + 63) & ~63;
  const int nkt = (a.Lk + KT - 1) / KT;
  walk_load_kv<NT>(a, a.K + b * a.k_bs + h * HD, a.V + b * a.v_bs + h * HD, smem, smem + WALK_BYTES, nkt, tid);
  __syncthreads();
  const uint32_t kbase = lds_addr(smem), vbase = kbase + WALK_BYTES;
  for (int base = wave * 16; base < a.Lq; base += QW * 16) {
    const bool qok = base + fi < a.Lq;
    const int q1[1] = {min(base + fi, a.Lq - 1)};
    const int q = q1[0];
    bf16x8 qf[2], dof[2];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(a.Q + b * a.q_bs + (long)q * a.q_rs + h * HD + ks * 32 + g * 8);
      dof[ks] = *reinterpret_cast<const bf16x8*>(a.dO + b * a.do_bs + (long)q * a.do_rs + h * HD + ks * 32 + g * 8);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.O + b * a.o_bs + (long)q * a.o_rs + h * HD + ks * 32 + g * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += bf2f((bf16_t)dof[ks][e]) * bf2f((bf16_t)of[e]);
    }
    const float delta = group_sum(dl);
    const float lse = a.LSE[((long)b * a.H + h) * a.Lq + q];
    if (qok && g == 0) a.Delta[((long)b * a.H + h) * a.Lq + q] = delta;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto tile = [&](int kt, auto full_) {          // FULL: see attn_fwd_kernel
      constexpr bool FULL = decltype(full_)::value;
      const uint32_t ktile = kbase + kt * KT * 128, vtile = vbase + kt * KT * 128;
      const int nsub = FULL ? 4 : min(4, (a.Lk - kt * KT + 15) >> 4);
      float4 bbv[1][4], mmv[4];
      load_bias_mask<1, BL2>(a, h, b, q1, kt * KT, g, nsub, bbv, mmv);
      f32x4 ds[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (nt >= nsub) { ds[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(ktile, nt * 16 + fi, ks * 4 + g), qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(vtile, nt * 16 + fi, ks * 4 + g), dof[ks], dp, 0, 0, 0);
        }
        const int key0 = kt * KT + nt * 16 + g * 4;
        s = apply_bias_mask<FULL, BL2>(s, bbv[0][nt], mmv[nt], key0, a.Lk, sc2);
        if (drop_.thr16) {
          float dm[4];
          drop_mul4(drop_, (uint32_t)(((long)b * a.H + h) * a.Lq + q) * (uint32_t)lkp + (uint32_t)key0, dm);
          dp[0] *= dm[0]; dp[1] *= dm[1]; dp[2] *= dm[2]; dp[3] *= dm[3];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[nt][r] = fast_exp2(s[r] - lse) * (dp[r] - delta);
        if (a.dS && qok && key0 < a.ds_ld)
          *reinterpret_cast<u32x2*>(a.dS + (((long)b * a.H + h) * a.Lq + q) * a.ds_ld + key0) =
              u32x2{pack_bf16(ds[nt][0], ds[nt][1]), pack_bf16(ds[nt][2], ds[nt][3])};
      }
      const bf16x8 dsf[2] = {pack8(ds[0], ds[1]), pack8(ds[2], ds[3])};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          if (2 * s2 >= nsub) continue;
          dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(ktile, s2, dt, lane), dsf[s2], dq[dt], 0, 0, 0);
        }
    };
    for (int kt = 0; kt < nkt; ++kt) tile(kt, PartTile{});          // general form only: see attn_bwd_dq_kernel
    if (qok) {
      bf16_t* op = a.dQ + b * a.dq_bs + (long)q * a.dq_rs + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<u32x2*>(op + dt * 16) = u32x2{pack_bf16(dq[dt][0] * a.scale, dq[dt][1] * a.scale),
                                                         pack_bf16(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// KW waves, KG groups of 16 keys per wave; the workgroup walks every (sequence using this K/V batch, 64-query tile).
template <int KW, int KG, bool RES, int NS = (RES ? 4 : 2), int WPS = ((KW == 4 && !RES) ? 3 : 4), bool BL2 = false>
// (second launch bound = waves per SIMD the register allocation must allow: these kernels hide their load -> MFMA -> exp
// chains only behind other waves, and left alone hipcc spends 170-230 VGPRs on the short-sequence forms (2 waves per SIMD);
// capped at 128 they run 1.3-1.5x faster.  The streamed 4-wave form needs more than 128: 35 spills under the cap.)
__global__ __launch_bounds__(64 * KW, WPS) void attn_bwd_dkv_kernel(AttnArgs a) {
  const DropSpec drop_ = drop_at_epoch(a.drop, a.drop_epoch);
  constexpr int NT = 64 * KW;
  __shared__ __attribute__((aligned(16))) char smem[NS][2 * KT * 128 + 2 * KT * 4];   // {Q tile, dO tile, LSE[64], Delta[64]} per slot
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  int bx_, h, bk;
  if (!attn_block(a, bx_, h, bk)) return;
  const float sc2 = a.scale * LOG2E;
  const int lkp = (a.Lk + 63) & ~63;
  // LEAN: score arithmetic without validity selects, dropout a compile-time property of the tile (see the tile).  Only the
  // 8-wave resident form (vision, N = 197) takes it: measured r03v, the straight-line code costs the streaming kernels
  // registers they do not have (N = 577: 41 scratch accesses inside the loop, 516 -> 712 us) and the two-wave text kernels
  // their latency hiding (30 -> 35 us).
  constexpr bool LEAN = RES && KW == 8;

  int key[KG]; bool kok[KG];
  bf16x8 kf[KG][2], vf[KG][2];
  f32x4 dk[KG][4], dv[KG][4];
#pragma unroll
  for (int gk = 0; gk < KG; ++gk) {
    const int k0 = (bx_ * KW * KG + wave * KG + gk) * 16;
    kok[gk] = k0 + fi < a.Lk;
    key[gk] = min(k0 + fi, a.Lk - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[gk][ks] = *reinterpret_cast<const bf16x8*>(a.K + bk * a.k_bs + (long)key[gk] * a.k_rs + h * HD + ks * 32 + g * 8);
      vf[gk][ks] = *reinterpret_cast<const bf16x8*>(a.V + bk * a.v_bs + (long)key[gk] * a.v_rs + h * HD + ks * 32 + g * 8);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[gk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[gk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }

  const int sb = a.seq_off ? a.seq_off[bk] : bk, se = a.seq_off ? a.seq_off[bk + 1] : bk + 1;
  const int nqt = (a.Lq + KT - 1) / KT;
  const int nit = (se - sb) * nqt;                   // flattened (sequence, q-tile) iterations
  if (nit > 0) {
    TileRegs<NT> rq, rdo;
    float rl = 0.f, rd = 0.f;
    auto fetch = [&](int it) {
      const int si = sb + it / nqt, qt = it % nqt;
      const int bb = a.seq_ids ? a.seq_ids[si] : si;
      tile_load<NT>(rq, a.Q + bb * a.q_bs + h * HD, a.q_rs, qt * KT, a.Lq, tid);
      tile_load<NT>(rdo, a.dO + bb * a.do_bs + h * HD, a.do_rs, qt * KT, a.Lq, tid);
      if (tid < KT) {
        // queries past Lq: LSE = +1e30 makes their P (and dS) exactly 0 in the tile arithmetic below, which therefore needs
        // no per-score validity select (the rows of Q / dO they multiply are clamped copies: finite)
        const int qq = min(qt * KT + tid, a.Lq - 1);
        const bool qv = qt * KT + tid < a.Lq;
        rl = qv ? a.LSE[((long)bb * a.H + h) * a.Lq + qq] : 1e30f;
        rd = qv ? a.Delta[((long)bb * a.H + h) * a.Lq + qq] : 0.f;
      }
    };
    auto commit = [&](int buf) {
      tile_store<NT>(rq, smem[buf], tid);
      tile_store<NT>(rdo, smem[buf] + KT * 128, tid);
      if (tid < KT) {
        reinterpret_cast<float*>(smem[buf] + 2 * KT * 128)[tid] = rl;
        reinterpret_cast<float*>(smem[buf] + 2 * KT * 128)[KT + tid] = rd;
      }
    };
    if (RES) {
      // (<= 4 query tiles) all requested before the first LDS write, as in the forward kernel
      TileRegs<NT> rqa[NS], rdoa[NS];
      float rla[NS], rda[NS];
#pragma unroll
      for (int it = 0; it < NS; ++it)
        if (it < nit) { fetch(it); rqa[it] = rq; rdoa[it] = rdo; rla[it] = rl; rda[it] = rd; }
#pragma unroll
      for (int it = 0; it < NS; ++it)
        if (it < nit) { rq = rqa[it]; rdo = rdoa[it]; rl = rla[it]; rd = rda[it]; commit(it); }
    } else {
      fetch(0);
      commit(0);
    }
    __syncthreads();
    const bool idle = (bx_ * KW * KG + wave * KG) * 16 >= a.Lk;      // all keys of this wave are padding
    if (RES && idle) return;
    // FULL (compile time): all 64 queries of the tile and all keys of this wave exist - the sub-tile tests and the validity
    // selects fold away and the tile is straight-line code (see attn_fwd_kernel)
    const bool keys_full = (bx_ * KW * KG + wave * KG + KG) * 16 <= a.Lk;
    auto tile = [&](int it, auto full_, auto drop_tag) {
      constexpr bool FULL = decltype(full_)::value;
      constexpr bool DROP = decltype(drop_tag)::value;          // LEAN only: probability dropout on, compile-time inside the tile
      const int si = sb + it / nqt, qt = it % nqt;
      const int nsub = FULL ? 4 : (idle ? 0 : min(4, (a.Lq - qt * KT + 15) >> 4));      // valid 16-query sub-tiles of this tile
      const int b = a.seq_ids ? a.seq_ids[si] : si;
      char* buf = smem[RES ? it : (it & 1)];
      const uint32_t qtile = lds_addr(buf), dotile = qtile + KT * 128;
      const float* lse_s = reinterpret_cast<const float*>(buf + 2 * KT * 128);
      const float* del_s = lse_s + KT;
      if (!RES && it + 1 < nit) fetch(it + 1);
      float mk[KG];
#pragma unroll
      for (int gk = 0; gk < KG; ++gk) mk[gk] = (a.mask ? a.mask[(long)b * a.mask_ld + key[gk]] : 0.f) * LOG2E;
      float4 btv[KG][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int gk = 0; gk < KG; ++gk) btv[gk][t] = float4{0.f, 0.f, 0.f, 0.f};
      if (a.biasT) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int gk = 0; gk < KG; ++gk)
            if (t < nsub) btv[gk][t] = *reinterpret_cast<const float4*>(a.biasT + ((long)h * a.Lk + key[gk]) * a.biasT_ld + qt * KT + t * 16 + g * 4);
        if (!BL2 && (a.dbg & 16)) {      // log2-unit bias in a kernel without the one-fma form: see load_bias_mask
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int gk = 0; gk < KG; ++gk) { btv[gk][t].x *= LN2; btv[gk][t].y *= LN2; btv[gk][t].z *= LN2; btv[gk][t].w *= LN2; }
        }
      }
      f32x4 p[KG][4], ds[KG][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t >= nsub) {
#pragma unroll
          for (int gk = 0; gk < KG; ++gk) { p[gk][t] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[gk][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
          continue;
        }
        f32x4 s[KG], dp[KG];
#pragma unroll
        for (int gk = 0; gk < KG; ++gk) { s[gk] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[gk] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 qfr = frag_rows(qtile, t * 16 + fi, ks * 4 + g), dofr = frag_rows(dotile, t * 16 + fi, ks * 4 + g);
#pragma unroll
          for (int gk = 0; gk < KG; ++gk) {
            s[gk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[gk][ks], s[gk], 0, 0, 0);
            dp[gk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofr, vf[gk][ks], dp[gk], 0, 0, 0);
          }
        }
        // this lane: keys = its own (one per group), queries qq0 .. qq0+3
        const int qq0 = qt * KT + t * 16 + g * 4;
        const float4 ls = *reinterpret_cast<const float4*>(lse_s + t * 16 + g * 4);
        const float4 dl = *reinterpret_cast<const float4*>(del_s + t * 16 + g * 4);
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
        for (int gk = 0; gk < KG; ++gk) {
          const float4 bb = btv[gk][t];
          const float bbv[4] = {bb.x, bb.y, bb.z, bb.w};
          if constexpr (LEAN) {
            // Branch-free: a key past Lk needs no masking here - column `key` of P / dS reaches only row `key` of dK / dV,
            // which is not stored - and a query past Lq has LSE = +1e30 (P = dS = 0).  The min keeps whatever sits in the
            // pad columns of the bias (undefined, possibly NaN: v_min returns the other operand) from reaching the exponent.
            const uint32_t e0 = (uint32_t)(((long)b * a.H + h) * a.Lq + qq0) * (uint32_t)lkp + (uint32_t)key[gk];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float x = BL2 ? fmaf(s[gk][r], sc2, bbv[r]) : fmaf(s[gk][r], sc2, fmaf(bbv[r], LOG2E, mk[gk]));
              const float pv = fast_exp2(fminf(x, 1e29f) - lsv[r]);
              const float dm = DROP ? drop_mul(drop_, e0 + (uint32_t)r * (uint32_t)lkp) : 1.f;
              p[gk][t][r] = DROP ? pv * dm : pv;
              ds[gk][t][r] = pv * ((DROP ? dp[gk][r] * dm : dp[gk][r]) - dlv[r]);
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool ok = FULL || (kok[gk] && (qq0 + r < a.Lq));
              const float pv = ok ? fast_exp2(BL2 ? fmaf(s[gk][r], sc2, bbv[r]) - lsv[r] : s[gk][r] * sc2 + bbv[r] * LOG2E + mk[gk] - lsv[r]) : 0.f;
              float dm = 1.f;
              if (drop_.thr16)
                dm = drop_mul(drop_, (uint32_t)(((long)b * a.H + h) * a.Lq + min(qq0 + r, a.Lq - 1)) * (uint32_t)lkp + (uint32_t)key[gk]);
              p[gk][t][r] = pv * dm;
              ds[gk][t][r] = pv * (dp[gk][r] * dm - dlv[r]);
            }
          }
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf[KG], dsf[KG];
#pragma unroll
        for (int gk = 0; gk < KG; ++gk) { pf[gk] = pack8(p[gk][2 * s2], p[gk][2 * s2 + 1]); dsf[gk] = pack8(ds[gk][2 * s2], ds[gk][2 * s2 + 1]); }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          if (2 * s2 >= nsub) continue;
          const bf16x8 dotr = frag_cols(dotile, s2, dt, lane), qtr = frag_cols(qtile, s2, dt, lane);
#pragma unroll
          for (int gk = 0; gk < KG; ++gk) {
            dv[gk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotr, pf[gk], dv[gk][dt], 0, 0, 0);
            dk[gk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr, dsf[gk], dk[gk][dt], 0, 0, 0);
          }
        }
      }
      if (!RES) {
        if (it + 1 < nit) commit((it + 1) & 1);
        __syncthreads();
      }
    };
    (void)keys_full;
    for (int it = 0; it < nit; ++it) {         // general (PartTile) form only: see attn_bwd_dq_kernel
      if (!LEAN || drop_.thr16) tile(it, PartTile{}, std::true_type{});
      else tile(it, PartTile{}, std::false_type{});
    }
  }
#pragma unroll
  for (int gk = 0; gk < KG; ++gk) {
    if (!kok[gk]) continue;
    bf16_t* kp = a.dK + bk * a.dk_bs + (long)key[gk] * a.dk_rs + h * HD + g * 4;
    bf16_t* vp = a.dV + bk * a.dv_bs + (long)key[gk] * a.dv_rs + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<u32x2*>(kp + dt * 16) = u32x2{pack_bf16(dk[gk][dt][0] * a.scale, dk[gk][dt][1] * a.scale),
                                                       pack_bf16(dk[gk][dt][2] * a.scale, dk[gk][dt][3] * a.scale)};
      *reinterpret_cast<u32x2*>(vp + dt * 16) = u32x2{pack_bf16(dv[gk][dt][0], dv[gk][dt][1]), pack_bf16(dv[gk][dt][2], dv[gk][dt][3])};
    }
  }
}

// ------------------------------------------------------------------------------------------ backward in one pass
// Self-attention with 64 < Lq, Lk <= 208 and no K/V sharing (the BEiT-2 blocks at N = 197): ONE eight-wave workgroup per
// (sequence, head) forms S, P, dP and dS once - 5 matrix products instead of the 7 of the dQ + dK/dV pair above, half the score
// arithmetic, Q / dO / K / V fetched once instead of three times.
//   phase A (key side, the tile arithmetic of attn_bwd_dkv_kernel): a wave owns the key strips {wave, wave + 8} (16 keys each, K / V
//     fragments in registers), walks the query strips in pairs (32 queries = one MFMA contraction) over Q and dO resident in LDS,
//     accumulates dK / dV in registers and leaves every dS tile in LDS as bf16, laid out [query strip][key][16 queries] - the
//     lane that owns a key writes its 4 queries as one 8-byte store, the wave 512 contiguous bytes;
//   phase B (query side): the K fragments go from registers into the LDS image Q occupied, and a wave forms dQ^T = K^T . dS^T for the
//     query strips {wave, wave + 8}: both operands by transposing reads (ds_read_b64_tr_b16), the dS fragment it has just read
//     is also what goes to the HBM dS stream (bias gradient), in the layout the dQ kernels write.
// Delta = rowsum(dO * O) is computed while Q / dO are staged (and still written out: callers keep the buffer).
// LDS image (140.6 KB, one workgroup per CU): dO | Q (later K) | 16 zero rows | dS blocks | 16 zero rows | LSE | Delta.  The order
// matters: a transposing fragment read of the last 32-row step runs 16 rows past its operand, and what it finds there must be
// finite AND meet a zero on the other side of the product - dO runs into Q (finite; its P is 0: no such queries), Q / K into
// the zero rows, a dS block into the next block (finite) while K has already run into the zero rows.
#define OP_ROWS 208
#define OP_REG (OP_ROWS * 128)
#define OP_BLK (OP_ROWS * 32)
#define OP_NBLK (OP_ROWS / 16)
#define OP_DO 0
#define OP_QK OP_REG
#define OP_Z1 (2 * OP_REG)
#define OP_DS (OP_Z1 + 16 * 128)
#define OP_Z2 (OP_DS + OP_NBLK * OP_BLK)
#define OP_LSE (OP_Z2 + 16 * 32)
#define OP_DEL (OP_LSE + OP_ROWS * 4)
#define OP_CS (OP_DEL + OP_ROWS * 4)          // [2 kinds: dQ, dV][8 waves][64 d] fp32 column sums of the wave's rows (colsum_ws)
#define OP_BYTES (OP_CS + 2 * 8 * 64 * 4)
// column sums over the 16 lanes that share g (rows fi = 0 .. 15 of a strip): afterwards lane fi == 0 of every g holds the 16 sums of its d = dt * 16 + 4 g + r
__device__ __forceinline__ void strip_colsum16(float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] += __shfl_xor(v[i], 1, 64); v[i] += __shfl_xor(v[i], 2, 64); v[i] += __shfl_xor(v[i], 4, 64); v[i] += __shfl_xor(v[i], 8, 64);
  }
}
template <bool BL2>
__global__ __launch_bounds__(512, 2) void attn_bwd_onepass_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[OP_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  int bx_, h, b;
  if (!attn_block(a, bx_, h, b)) return;
  const float sc2 = a.scale * LOG2E;
  const int nsq = (a.Lq + 15) >> 4, nsk = (a.Lk + 15) >> 4;      // 16-row strips of queries / keys
  const uint32_t lbase = lds_addr(smem), dotile = lbase + OP_DO, qtile = lbase + OP_QK;
  float* lse_s = reinterpret_cast<float*>(smem + OP_LSE);
  float* del_s = reinterpret_cast<float*>(smem + OP_DEL);

  // this wave's key strips (wave, wave + 8): K / V rows straight into MFMA fragments
  int key[2]; bool kok[2];
  bf16x8 kf[2][2], vf[2][2];
#pragma unroll
  for (int gk = 0; gk < 2; ++gk) {
    const int k0 = (wave + 8 * gk) * 16;
    kok[gk] = k0 + fi < a.Lk;
    key[gk] = min(k0 + fi, a.Lk - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[gk][ks] = *reinterpret_cast<const bf16x8*>(a.K + b * a.k_bs + (long)key[gk] * a.k_rs + h * HD + ks * 32 + g * 8);
      vf[gk][ks] = *reinterpret_cast<const bf16x8*>(a.V + b * a.v_bs + (long)key[gk] * a.v_rs + h * HD + ks * 32 + g * 8);
    }
  }
  // zero rows and dS blocks (strips no wave owns, queries past Lq and the columns of keys past Lk stay zero)
  for (int o = tid * 16; o < OP_LSE - OP_Z1; o += 512 * 16) *reinterpret_cast<u32x4*>(smem + OP_Z1 + o) = u32x4{0u, 0u, 0u, 0u};
  {
    // Q and dO -> LDS (rows past Lq: copies of the last row - finite, and their LSE = +1e30 makes P = dS = 0), Delta on the way
    u32x4 rq[4], rdo[4], ro[4];
    float rl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 512;
      if (c < OP_ROWS * 8) {
        const int row = c >> 3, ch = c & 7, gr = min(row, a.Lq - 1);
        rq[i] = *reinterpret_cast<const u32x4*>(a.Q + b * a.q_bs + (long)gr * a.q_rs + h * HD + ch * 8);
        rdo[i] = *reinterpret_cast<const u32x4*>(a.dO + b * a.do_bs + (long)gr * a.do_rs + h * HD + ch * 8);
        ro[i] = *reinterpret_cast<const u32x4*>(a.O + b * a.o_bs + (long)gr * a.o_rs + h * HD + ch * 8);
        rl[i] = (ch == 0 && row < a.Lq) ? a.LSE[((long)b * a.H + h) * a.Lq + row] : 1e30f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 512;
      if (c < OP_ROWS * 8) {            // wave-uniform: 1664 = 26 waves of chunks
        const int row = c >> 3, ch = c & 7;
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) dl += bf_lo(rdo[i][e]) * bf_lo(ro[i][e]) + bf_hi(rdo[i][e]) * bf_hi(ro[i][e]);
        dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64); dl += __shfl_xor(dl, 4, 64);
        const int off = row * 128 + ((ch ^ (row & 7)) << 4);
        *reinterpret_cast<u32x4*>(smem + OP_QK + off) = rq[i];
        *reinterpret_cast<u32x4*>(smem + OP_DO + off) = rdo[i];
        if (ch == 0) {
          const bool ok = row < a.Lq;
          lse_s[row] = rl[i];
          del_s[row] = ok ? dl : 0.f;
          if (ok) a.Delta[((long)b * a.H + h) * a.Lq + row] = dl;
        }
      }
    }
  }
  __syncthreads();

  // ---- phase A
  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int gk = 0; gk < 2; ++gk)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[gk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[gk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto phase_a = [&](auto ng_) {
    constexpr int NG = decltype(ng_)::value;
    float mk[NG];
#pragma unroll
    for (int gk = 0; gk < NG; ++gk) mk[gk] = (a.mask ? a.mask[(long)b * a.mask_ld + key[gk]] : 0.f) * LOG2E;
    // relative-position bias of this lane's keys for one pair of query strips (4 queries per strip and lane), fetched ONE PAIR
    // AHEAD: left to the top of the pair that uses them the loads sit exposed in front of the exponentials (two waves per SIMD
    // do not cover an L2 round trip with eight MFMAs).  Columns up to 32 * ceil(nsq / 2) - 1 < biasT_ld (a multiple of 64 >= Lq).
    const int last_pair = ((nsq + 1) >> 1) - 1;
    float4 bnext[NG][2];
    auto load_bias = [&](int s) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gk = 0; gk < NG; ++gk) {
          bnext[gk][t] = float4{0.f, 0.f, 0.f, 0.f};
          if (BL2 || a.biasT) bnext[gk][t] = *reinterpret_cast<const float4*>(a.biasT + ((long)h * a.Lk + key[gk]) * a.biasT_ld + 32 * s + 16 * t + g * 4);
        }
    };
    load_bias(0);
    auto pair = [&](int s, auto full_) {
      constexpr int NTQ = decltype(full_)::value ? 2 : 1;        // query strips of this pair that exist
      float4 btv[NG][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gk = 0; gk < NG; ++gk) {
          btv[gk][t] = bnext[gk][t];
          if (!BL2 && (a.dbg & 16)) { btv[gk][t].x *= LN2; btv[gk][t].y *= LN2; btv[gk][t].z *= LN2; btv[gk][t].w *= LN2; }
        }
      load_bias(min(s + 1, last_pair));
      f32x4 p[NG][2], ds[NG][2];
#pragma unroll
      for (int gk = 0; gk < NG; ++gk) { p[gk][1] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[gk][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int t = 0; t < NTQ; ++t) {
        f32x4 sa[NG], dp[NG];
#pragma unroll
        for (int gk = 0; gk < NG; ++gk) { sa[gk] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[gk] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 qfr = frag_rows(qtile, 32 * s + 16 * t + fi, ks * 4 + g), dofr = frag_rows(dotile, 32 * s + 16 * t + fi, ks * 4 + g);
#pragma unroll
          for (int gk = 0; gk < NG; ++gk) {
            sa[gk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[gk][ks], sa[gk], 0, 0, 0);
            dp[gk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofr, vf[gk][ks], dp[gk], 0, 0, 0);
          }
        }
        // this lane: its own key per strip, queries 32 s + 16 t + 4 g + r
        const float4 ls = *reinterpret_cast<const float4*>(lse_s + 32 * s + 16 * t + g * 4);
        const float4 dl = *reinterpret_cast<const float4*>(del_s + 32 * s + 16 * t + g * 4);
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
        for (int gk = 0; gk < NG; ++gk) {
          const float bbv[4] = {btv[gk][t].x, btv[gk][t].y, btv[gk][t].z, btv[gk][t].w};
          // branch-free as the LEAN tile of attn_bwd_dkv_kernel: a query past Lq has LSE = +1e30 (P = dS = 0); the min keeps the pad
          // columns of the bias (undefined) out of the exponent; a key past Lk reaches dK / dV rows that are not stored, and its
          // dS column is zeroed where it is written for phase B
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = BL2 ? fmaf(sa[gk][r], sc2, bbv[r]) : fmaf(sa[gk][r], sc2, fmaf(bbv[r], LOG2E, mk[gk]));
            const float pv = fast_exp2(fminf(x, 1e29f) - lsv[r]);
            p[gk][t][r] = pv;
            ds[gk][t][r] = pv * (dp[gk][r] - dlv[r]);
          }
        }
      }
      bf16x8 pf[NG], dsf[NG];
#pragma unroll
      for (int gk = 0; gk < NG; ++gk) {
        pf[gk] = pack8(p[gk][0], p[gk][1]);
        dsf[gk] = pack8(ds[gk][0], ds[gk][1]);
        const u32x4 w = __builtin_bit_cast(u32x4, dsf[gk]);
        char* blk = smem + OP_DS + (2 * s) * OP_BLK + ((wave + 8 * gk) * 16 + fi) * 32 + g * 8;
        *reinterpret_cast<u32x2*>(blk) = kok[gk] ? u32x2{w[0], w[1]} : u32x2{0u, 0u};
        if (NTQ == 2) *reinterpret_cast<u32x2*>(blk + OP_BLK) = kok[gk] ? u32x2{w[2], w[3]} : u32x2{0u, 0u};
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 dotr = frag_cols(dotile, s, dt, lane), qtr = frag_cols(qtile, s, dt, lane);
#pragma unroll
        for (int gk = 0; gk < NG; ++gk) {
          dv[gk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotr, pf[gk], dv[gk][dt], 0, 0, 0);
          dk[gk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr, dsf[gk], dk[gk][dt], 0, 0, 0);
        }
      }
    };
    for (int s = 0; s < (nsq >> 1); ++s) pair(s, std::true_type{});
    if (nsq & 1) pair(nsq >> 1, std::false_type{});
    float vs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) vs[i] = 0.f;
#pragma unroll
    for (int gk = 0; gk < NG; ++gk) {
      if (!kok[gk]) continue;
      bf16_t* kp = a.dK + b * a.dk_bs + (long)key[gk] * a.dk_rs + h * HD + g * 4;
      bf16_t* vp = a.dV + b * a.dv_bs + (long)key[gk] * a.dv_rs + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<u32x2*>(kp + dt * 16) = u32x2{pack_bf16(dk[gk][dt][0] * a.scale, dk[gk][dt][1] * a.scale),
                                                         pack_bf16(dk[gk][dt][2] * a.scale, dk[gk][dt][3] * a.scale)};
        const u32x2 wv = u32x2{pack_bf16(dv[gk][dt][0], dv[gk][dt][1]), pack_bf16(dv[gk][dt][2], dv[gk][dt][3])};
        *reinterpret_cast<u32x2*>(vp + dt * 16) = wv;
        vs[dt * 4 + 0] += bf_lo(wv[0]); vs[dt * 4 + 1] += bf_hi(wv[0]); vs[dt * 4 + 2] += bf_lo(wv[1]); vs[dt * 4 + 3] += bf_hi(wv[1]);
      }
    }
    if (a.colsum_ws) {                 // this wave's share of colsum(dV): keys past Lk added nothing (skipped above)
      strip_colsum16(vs);
      if (fi == 0) {
        float* cs = reinterpret_cast<float*>(smem + OP_CS) + (8 + wave) * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[dt * 16 + g * 4 + r] = vs[dt * 4 + r];
      }
    }
  };
  const int ng = wave + 8 < nsk ? 2 : (wave < nsk ? 1 : 0);
  if (a.colsum_ws && lane < 64) {      // zero this wave's slots first: waves without key / query strips contribute zeros
    float* cs = reinterpret_cast<float*>(smem + OP_CS);
    cs[wave * 64 + lane] = 0.f; cs[(8 + wave) * 64 + lane] = 0.f;
  }
  if (ng == 2) phase_a(std::integral_constant<int, 2>{});
  else if (ng == 1) phase_a(std::integral_constant<int, 1>{});
  __syncthreads();                     // every wave is done with Q
  // K rows of the strips this wave holds -> the image Q occupied (strips nobody holds keep Q rows: finite, and their dS is zero)
#pragma unroll
  for (int gk = 0; gk < 2; ++gk)
    if (gk < ng) {
      const int row = (wave + 8 * gk) * 16 + fi;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        *reinterpret_cast<bf16x8*>(smem + OP_QK + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4)) = kf[gk][ks];
    }
  __syncthreads();

  // ---- phase B
  auto phase_b = [&](auto nq_) {
    constexpr int NQ = decltype(nq_)::value;
    f32x4 dq[NQ][4];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r4 = fi >> 2, c4 = fi & 3;
    const int kend = min(nsk * 16, a.ds_ld);
    for (int s = 0; s < ((nsk + 1) >> 1); ++s) {
      bf16x8 dsf[NQ];
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const uint32_t a0 = lbase + OP_DS + (wave + 8 * j) * OP_BLK + (32 * s + 4 * g + r4) * 32 + c4 * 8;
        dsf[j] = lds_read_tr_frag(a0, a0 + 16 * 32);
        const int q = (wave + 8 * j) * 16 + fi;
        if (a.dS && q < a.Lq) {          // the dS stream of the bias gradient: [B][H][Lq][ds_ld], this lane = one query, 2 x 4 keys.
                                         // Written: columns < min(16 ceil(Lk / 16), ds_ld) (zeros past Lk); the two-kernel form fills all ds_ld.
                                         // Contract (include/x2vlm_hip.h, x2_relpos_bias_bwd): consumers read columns < 8 ceil(Lk / 8) only.
          const u32x4 w = __builtin_bit_cast(u32x4, dsf[j]);
          bf16_t* dsp = a.dS + (((long)b * a.H + h) * a.Lq + q) * a.ds_ld + 32 * s + 4 * g;
          if (32 * s + 4 * g < kend) *reinterpret_cast<u32x2*>(dsp) = u32x2{w[0], w[1]};
          if (32 * s + 16 + 4 * g < kend) *reinterpret_cast<u32x2*>(dsp + 16) = u32x2{w[2], w[3]};
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 ktr = frag_cols(qtile, s, dt, lane);
#pragma unroll
        for (int j = 0; j < NQ; ++j) dq[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr, dsf[j], dq[j][dt], 0, 0, 0);
      }
    }
    float qs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) qs[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = (wave + 8 * j) * 16 + fi;
      if (q >= a.Lq) continue;
      bf16_t* op = a.dQ + b * a.dq_bs + (long)q * a.dq_rs + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const u32x2 wq = u32x2{pack_bf16(dq[j][dt][0] * a.scale, dq[j][dt][1] * a.scale), pack_bf16(dq[j][dt][2] * a.scale, dq[j][dt][3] * a.scale)};
        *reinterpret_cast<u32x2*>(op + dt * 16) = wq;
        qs[dt * 4 + 0] += bf_lo(wq[0]); qs[dt * 4 + 1] += bf_hi(wq[0]); qs[dt * 4 + 2] += bf_lo(wq[1]); qs[dt * 4 + 3] += bf_hi(wq[1]);
      }
    }
    if (a.colsum_ws) {
      strip_colsum16(qs);
      if (fi == 0) {
        float* cs = reinterpret_cast<float*>(smem + OP_CS) + wave * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) cs[dt * 16 + g * 4 + r] = qs[dt * 4 + r];
      }
    }
  };
  if (wave + 8 < nsq) phase_b(std::integral_constant<int, 2>{});
  else if (wave < nsq) phase_b(std::integral_constant<int, 1>{});
  if (a.colsum_ws) {                   // (block-uniform) the eight waves' sums in wave order -> this (sequence, head)'s 2 x 64 partial columns
    __syncthreads();
    if (tid < 128) {
      const int kind = tid >> 6, d = tid & 63;
      const float* cs = reinterpret_cast<const float*>(smem + OP_CS) + kind * 8 * 64 + d;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += cs[w * 64];
      a.colsum_ws[((long)b * 2 + kind) * (a.H * HD) + h * HD + d] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------ backward in one pass, shared K/V
// Cross-attention of the fusion stack (text rows -> the image tokens several of them share; Lq <= 128, Lk <= 208, no bias): the
// one-pass scheme above with ONE eight-wave workgroup per (image, head) that walks the sequences using the image (CSR seq_off /
// seq_ids) in chunks of 8 query strips.  A sequence is padded to whole 16-query strips (30 tokens -> 2 strips, so a chunk is 4
// sequences and a 32-query contraction step is one sequence): every strip belongs to one sequence, its additive key mask and
// its dropout element base are per-strip values, and the pad queries carry LSE = +1e30 (P = dS = 0).  dK / dV stay in the
// accumulators across the chunks; K sits in an LDS image of its own (phase B of one chunk, phase A of the next need K and Q).
//   per chunk: stage Q / dO / LSE / Delta / mask rows -> barrier -> phase A -> barrier -> phase B (one strip per wave) -> barrier
// Replaces attn_bwd_dq_grouped_kernel + attn_bwd_dkv_kernel<4, 1, false> (which walked 64-row tiles holding ONE 30-row
// sequence each, one barrier and one exposed load per sequence): 5 matrix products instead of 7, and the K/V-side gradients no
// longer need a stream of their own to hide behind.
#define XP_STRIPS 8
#define XP_ROWS (XP_STRIPS * 16)
#define XP_DO 0
#define XP_Q (XP_ROWS * 128)
#define XP_Z1 (2 * XP_ROWS * 128)
#define XP_K (XP_Z1 + 16 * 128)
#define XP_Z3 (XP_K + OP_REG)
#define XP_DS (XP_Z3 + 16 * 128)
#define XP_Z2 (XP_DS + XP_STRIPS * OP_BLK)
#define XP_LSE (XP_Z2 + 16 * 32)
#define XP_DEL (XP_LSE + XP_ROWS * 4)
#define XP_MASK (XP_DEL + XP_ROWS * 4)                 // [8 strips][208 keys] fp32, x log2 e
#define XP_SB (XP_MASK + XP_STRIPS * OP_ROWS * 4)      // [8] sequence of every strip (-1: none)
#define XP_BYTES (XP_SB + 64)
template <bool DROP>
__global__ __launch_bounds__(512, 2) void attn_bwd_onepass_grouped_kernel(AttnArgs a) {
  const DropSpec drop_ = drop_at_epoch(a.drop, a.drop_epoch);
  __shared__ __attribute__((aligned(16))) char smem[XP_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, bk = blockIdx.z;
  const float sc2 = a.scale * LOG2E;
  const int lkp = (a.Lk + 63) & ~63;
  const int nsk = (a.Lk + 15) >> 4;                    // key strips (<= 13)
  const int sps = (a.Lq + 15) >> 4;                    // strips per sequence (<= 8)
  const int spc = XP_STRIPS / sps;                     // sequences per chunk
  const int sb = a.seq_off[bk], nseq = a.seq_off[bk + 1] - sb;
  const uint32_t lbase = lds_addr(smem), dotile = lbase + XP_DO, qtile = lbase + XP_Q, ktile = lbase + XP_K;
  float* lse_s = reinterpret_cast<float*>(smem + XP_LSE);
  float* del_s = reinterpret_cast<float*>(smem + XP_DEL);
  float* mask_s = reinterpret_cast<float*>(smem + XP_MASK);
  int* sb_s = reinterpret_cast<int*>(smem + XP_SB);

  int key[2]; bool kok[2];
  bf16x8 kf[2][2], vf[2][2];
  const int ng = wave + 8 < nsk ? 2 : (wave < nsk ? 1 : 0);
#pragma unroll
  for (int gk = 0; gk < 2; ++gk) {
    const int k0 = (wave + 8 * gk) * 16;
    kok[gk] = k0 + fi < a.Lk;
    key[gk] = min(k0 + fi, a.Lk - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[gk][ks] = *reinterpret_cast<const bf16x8*>(a.K + bk * a.k_bs + (long)key[gk] * a.k_rs + h * HD + ks * 32 + g * 8);
      vf[gk][ks] = *reinterpret_cast<const bf16x8*>(a.V + bk * a.v_bs + (long)key[gk] * a.v_rs + h * HD + ks * 32 + g * 8);
    }
  }
  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int gk = 0; gk < 2; ++gk)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[gk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[gk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  if (nseq > 0) {                      // block-uniform
    // zero rows, dS blocks; K rows from the fragments (strips nobody holds: zeros - their dS rows are zero as well, but 0 x NaN is not)
    for (int o = tid * 16; o < 16 * 128; o += 512 * 16) *reinterpret_cast<u32x4*>(smem + XP_Z1 + o) = u32x4{0u, 0u, 0u, 0u};
    for (int o = tid * 16; o < XP_LSE - XP_Z3; o += 512 * 16) *reinterpret_cast<u32x4*>(smem + XP_Z3 + o) = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int gk = 0; gk < 2; ++gk) {
      const int row = (wave + 8 * gk) * 16 + fi;
      if (row < OP_ROWS) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          *reinterpret_cast<bf16x8*>(smem + XP_K + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4)) = gk < ng ? kf[gk][ks] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    const int nchunk = (nseq + spc - 1) / spc;
    for (int c = 0; c < nchunk; ++c) {
      const int nsc = min(spc, nseq - c * spc), nvs = nsc * sps;       // sequences / strips of this chunk
      {
        // stage the chunk: row = strip * 16 + r, strip -> (sequence slot strip / sps, query strip strip % sps)
        u32x4 rq[2], rdo[2], ro[2];
        float rl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int cc = tid + i * 512, row = cc >> 3, ch = cc & 7, strip = row >> 4, slot = strip / sps, q = (strip - slot * sps) * 16 + (row & 15);
          const bool ok = slot < nsc && q < a.Lq;
          rq[i] = rdo[i] = ro[i] = u32x4{0u, 0u, 0u, 0u};
          rl[i] = 1e30f;
          if (ok) {
            const int b = a.seq_ids[sb + c * spc + slot];
            rq[i] = *reinterpret_cast<const u32x4*>(a.Q + b * a.q_bs + (long)q * a.q_rs + h * HD + ch * 8);
            rdo[i] = *reinterpret_cast<const u32x4*>(a.dO + b * a.do_bs + (long)q * a.do_rs + h * HD + ch * 8);
            ro[i] = *reinterpret_cast<const u32x4*>(a.O + b * a.o_bs + (long)q * a.o_rs + h * HD + ch * 8);
            if (ch == 0) rl[i] = a.LSE[((long)b * a.H + h) * a.Lq + q];
          }
        }
        // additive key mask of every strip's sequence (x log2 e), and the sequence itself
        for (int e = tid; e < XP_STRIPS * OP_ROWS; e += 512) {
          const int strip = e / OP_ROWS, k = e - strip * OP_ROWS, slot = strip / sps;
          float m = 0.f;
          if (a.mask && slot < nsc && k < a.Lk) m = a.mask[(long)a.seq_ids[sb + c * spc + slot] * a.mask_ld + k] * LOG2E;
          mask_s[e] = m;
        }
        if (tid < XP_STRIPS) { const int slot = tid / sps; sb_s[tid] = slot < nsc ? a.seq_ids[sb + c * spc + slot] : -1; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int cc = tid + i * 512, row = cc >> 3, ch = cc & 7, strip = row >> 4, slot = strip / sps, q = (strip - slot * sps) * 16 + (row & 15);
          const bool ok = slot < nsc && q < a.Lq;
          float dl = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) dl += bf_lo(rdo[i][e]) * bf_lo(ro[i][e]) + bf_hi(rdo[i][e]) * bf_hi(ro[i][e]);
          dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64); dl += __shfl_xor(dl, 4, 64);
          const int off = row * 128 + ((ch ^ (row & 7)) << 4);
          *reinterpret_cast<u32x4*>(smem + XP_Q + off) = rq[i];
          *reinterpret_cast<u32x4*>(smem + XP_DO + off) = rdo[i];
          if (ch == 0) {
            lse_s[row] = rl[i];
            del_s[row] = ok ? dl : 0.f;
            if (ok) a.Delta[((long)a.seq_ids[sb + c * spc + slot] * a.H + h) * a.Lq + q] = dl;
          }
        }
      }
      __syncthreads();
      // ---- phase A
      auto phase_a = [&](auto ng_) {
        constexpr int NG = decltype(ng_)::value;
        auto pair = [&](int s, auto full_) {
          constexpr int NTQ = decltype(full_)::value ? 2 : 1;
          f32x4 p[NG][2], ds[NG][2];
#pragma unroll
          for (int gk = 0; gk < NG; ++gk) { p[gk][1] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[gk][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int t = 0; t < NTQ; ++t) {
            const int strip = 2 * s + t;
            f32x4 sa[NG], dp[NG];
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) { sa[gk] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[gk] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const bf16x8 qfr = frag_rows(qtile, strip * 16 + fi, ks * 4 + g), dofr = frag_rows(dotile, strip * 16 + fi, ks * 4 + g);
#pragma unroll
              for (int gk = 0; gk < NG; ++gk) {
                sa[gk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[gk][ks], sa[gk], 0, 0, 0);
                dp[gk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofr, vf[gk][ks], dp[gk], 0, 0, 0);
              }
            }
            const float4 ls = *reinterpret_cast<const float4*>(lse_s + strip * 16 + g * 4);
            const float4 dl = *reinterpret_cast<const float4*>(del_s + strip * 16 + g * 4);
            const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
            // dropout element of (query, key): ((b H + h) Lq + query) lkp + key; pad queries (P = 0 anyway) take the last row's
            const int slot = strip / sps, qq0 = (strip - slot * sps) * 16 + g * 4;
            uint32_t ebase = 0;
            if (DROP) ebase = (uint32_t)(((long)sb_s[strip] * a.H + h) * a.Lq);
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) {
              const float mk = mask_s[strip * OP_ROWS + key[gk]];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float pv = fast_exp2(fmaf(sa[gk][r], sc2, mk) - lsv[r]);
                float dm = 1.f;
                if (DROP) dm = drop_mul(drop_, (ebase + (uint32_t)min(qq0 + r, a.Lq - 1)) * (uint32_t)lkp + (uint32_t)key[gk]);
                p[gk][t][r] = DROP ? pv * dm : pv;
                ds[gk][t][r] = pv * ((DROP ? dp[gk][r] * dm : dp[gk][r]) - dlv[r]);
              }
            }
          }
          bf16x8 pf[NG], dsf[NG];
#pragma unroll
          for (int gk = 0; gk < NG; ++gk) {
            pf[gk] = pack8(p[gk][0], p[gk][1]);
            dsf[gk] = pack8(ds[gk][0], ds[gk][1]);
            const u32x4 w = __builtin_bit_cast(u32x4, dsf[gk]);
            char* blk = smem + XP_DS + (2 * s) * OP_BLK + ((wave + 8 * gk) * 16 + fi) * 32 + g * 8;
            *reinterpret_cast<u32x2*>(blk) = kok[gk] ? u32x2{w[0], w[1]} : u32x2{0u, 0u};
            if (NTQ == 2) *reinterpret_cast<u32x2*>(blk + OP_BLK) = kok[gk] ? u32x2{w[2], w[3]} : u32x2{0u, 0u};
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 dotr = frag_cols(dotile, s, dt, lane), qtr = frag_cols(qtile, s, dt, lane);
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) {
              dv[gk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotr, pf[gk], dv[gk][dt], 0, 0, 0);
              dk[gk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr, dsf[gk], dk[gk][dt], 0, 0, 0);
            }
          }
        };
        for (int s = 0; s < (nvs >> 1); ++s) pair(s, std::true_type{});
        if (nvs & 1) pair(nvs >> 1, std::false_type{});
      };
      if (ng == 2) phase_a(std::integral_constant<int, 2>{});
      else if (ng == 1) phase_a(std::integral_constant<int, 1>{});
      __syncthreads();
      // ---- phase B: strip `wave` of the chunk
      if (wave < nvs) {
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int r4 = fi >> 2, c4 = fi & 3;
        for (int s = 0; s < ((nsk + 1) >> 1); ++s) {
          const uint32_t a0 = lbase + XP_DS + wave * OP_BLK + (32 * s + 4 * g + r4) * 32 + c4 * 8;
          const bf16x8 dsf = lds_read_tr_frag(a0, a0 + 16 * 32);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(ktile, s, dt, lane), dsf, dq[dt], 0, 0, 0);
        }
        const int slot = wave / sps, q = (wave - slot * sps) * 16 + fi;
        if (q < a.Lq) {
          const int b = sb_s[wave];
          bf16_t* op = a.dQ + b * a.dq_bs + (long)q * a.dq_rs + h * HD + g * 4;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<u32x2*>(op + dt * 16) = u32x2{pack_bf16(dq[dt][0] * a.scale, dq[dt][1] * a.scale),
                                                             pack_bf16(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
        }
      }
      __syncthreads();                 // the next chunk overwrites Q / dO / the per-strip tables, and phase A the dS blocks
    }
  }
#pragma unroll
  for (int gk = 0; gk < 2; ++gk) {
    if (gk >= ng || !kok[gk]) continue;
    bf16_t* kp = a.dK + bk * a.dk_bs + (long)key[gk] * a.dk_rs + h * HD + g * 4;
    bf16_t* vp = a.dV + bk * a.dv_bs + (long)key[gk] * a.dv_rs + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<u32x2*>(kp + dt * 16) = u32x2{pack_bf16(dk[gk][dt][0] * a.scale, dk[gk][dt][1] * a.scale),
                                                       pack_bf16(dk[gk][dt][2] * a.scale, dk[gk][dt][3] * a.scale)};
      *reinterpret_cast<u32x2*>(vp + dt * 16) = u32x2{pack_bf16(dv[gk][dt][0], dv[gk][dt][1]), pack_bf16(dv[gk][dt][2], dv[gk][dt][3])};
    }
  }
}

// ------------------------------------------------------------------------------------------ backward in one pass, long sequences
// Self-attention with 208 < Lq <= 640, 208 < Lk <= 768 and no K/V sharing (the BEiT-2 blocks of X2VLM-large at 384 px: N = 577; the
// autograd backward of /root/reference/models/beit2.py:135-159 - q * scale, QK^T + relative_position_bias, softmax, PV): a
// (sequence, head) no longer fits a workgroup's registers + LDS at once, so ONE eight-wave workgroup per (sequence, head) walks it as
//   for every key PART (256 keys = 16 strips, two per wave: K / V fragments in registers, dK / dV accumulators, K rows in LDS)
//     for every query CHUNK (128 queries = 8 strips: Q / dO staged into LDS, Delta recomputed from dO . O on the way)
//       phase A: every wave forms S, P, dP, dS of its key strips x the chunk's query strips ONCE (the tile arithmetic of
//                attn_bwd_onepass_kernel), accumulates dK / dV and leaves dS in LDS as bf16 [query strip][key][16 queries]
//       phase B: wave w forms the chunk's query strip w: dQ += K^T . dS^T over the part's keys; the dS fragments it reads are also the
//                HBM dS stream of the bias gradient
// dQ of a strip is a sum over the parts: it travels between parts as an fp32 partial in the caller's workspace (`ws`: [B][H][chunks][8
// strips][4][64 lanes][4] floats - written and read back by the SAME lane, added in part order: deterministic, no atomics) and leaves as
// bf16 after the last part.  5 matrix products per score instead of the 7 of the dQ + dK/dV pair, scores exponentiated once.
// LDS (129 KB): dO chunk | Q chunk | K part (256 rows; strips the part does not have: zeros) | dS blocks (8 x 8 KB) | LSE | Delta.
#define LP_STRIPS 8
#define LP_ROWS (LP_STRIPS * 16)
#define LP_KEYS 256
#define LP_BLK (LP_KEYS * 32)
#define LP_DO 0
#define LP_Q (LP_ROWS * 128)
#define LP_K (2 * LP_ROWS * 128)
#define LP_DS (LP_K + LP_KEYS * 128)
#define LP_LSE (LP_DS + LP_STRIPS * LP_BLK)
#define LP_DEL (LP_LSE + LP_ROWS * 4)
#define LP_CS (LP_DEL + LP_ROWS * 4)           // [2 kinds: dQ, dV][8 waves][64 d] fp32 column sums (colsum_ws), accumulated over chunks / parts
#define LP_BYTES (LP_CS + 2 * 8 * 64 * 4)
#define LP_MAX_LQ 640
#define LP_MAX_LK 768
template <bool BL2>
__global__ __launch_bounds__(512, 2) void attn_bwd_onepass_long_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[LP_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  int bx_, h, b;
  if (!attn_block(a, bx_, h, b)) return;
  const float sc2 = a.scale * LOG2E;
  const int nsq = (a.Lq + 15) >> 4, nsk = (a.Lk + 15) >> 4;
  const int nchunk = (nsq + LP_STRIPS - 1) / LP_STRIPS, npart = (nsk + 15) >> 4;
  const uint32_t lbase = lds_addr(smem), dotile = lbase + LP_DO, qtile = lbase + LP_Q, ktile = lbase + LP_K;
  float* lse_s = reinterpret_cast<float*>(smem + LP_LSE);
  float* del_s = reinterpret_cast<float*>(smem + LP_DEL);
  float* wsq = a.ws + (((long)b * a.H + h) * nchunk * LP_STRIPS + wave) * 1024 + lane * 4;      // + chunk * 8192 + dt * 256
  const int r4 = fi >> 2, c4 = fi & 3;
  // column-sum slots of this wave (only this wave touches them until the end); addressed from a scalar wave number where they are used
#define LP_CS_SLOT(kind) (reinterpret_cast<float*>(smem + LP_CS) + ((kind) * 8 + __builtin_amdgcn_readfirstlane(wave)) * 64)
  if (a.colsum_ws) { LP_CS_SLOT(0)[lane] = 0.f; LP_CS_SLOT(1)[lane] = 0.f; }

  for (int part = 0; part < npart; ++part) {
    const int nkp = min(16, nsk - part * 16);                      // key strips of this part
    const int ng = wave + 8 < nkp ? 2 : (wave < nkp ? 1 : 0);
    int key[2]; bool kok[2];
    bf16x8 kf[2][2], vf[2][2];
#pragma unroll
    for (int gk = 0; gk < 2; ++gk) {
      const int k0 = (part * 16 + wave + 8 * gk) * 16;
      kok[gk] = k0 + fi < a.Lk;
      key[gk] = min(k0 + fi, a.Lk - 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[gk][ks] = *reinterpret_cast<const bf16x8*>(a.K + b * a.k_bs + (long)key[gk] * a.k_rs + h * HD + ks * 32 + g * 8);
        vf[gk][ks] = *reinterpret_cast<const bf16x8*>(a.V + b * a.v_bs + (long)key[gk] * a.v_rs + h * HD + ks * 32 + g * 8);
      }
    }
    // (the barrier that ended the previous part's last chunk: nobody reads the K image or the dS blocks any more)
#pragma unroll
    for (int gk = 0; gk < 2; ++gk) {
      const int row = (wave + 8 * gk) * 16 + fi;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        *reinterpret_cast<bf16x8*>(smem + LP_K + row * 128 + (((ks * 4 + g) ^ (row & 7)) << 4)) = gk < ng ? kf[gk][ks] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    if (nkp < 16)        // block-uniform: key rows no wave writes in this part still hold the previous part's dS
      for (int o = tid * 16; o < LP_STRIPS * LP_BLK; o += 512 * 16) *reinterpret_cast<u32x4*>(smem + LP_DS + o) = u32x4{0u, 0u, 0u, 0u};
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int gk = 0; gk < 2; ++gk)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dk[gk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[gk][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int c = 0; c < nchunk; ++c) {
      const int nvs = min(LP_STRIPS, nsq - c * LP_STRIPS);         // query strips of this chunk
      const int q0c = c * LP_ROWS;
      {
        // stage the chunk (rows past Lq: copies of the last row - finite - with LSE = +1e30: P = dS = 0), Delta on the way
        u32x4 rq[2], rdo[2], ro[2];
        float rl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int cc = tid + i * 512, row = cc >> 3, ch = cc & 7, q = q0c + row, gr = min(q, a.Lq - 1);
          rq[i] = *reinterpret_cast<const u32x4*>(a.Q + b * a.q_bs + (long)gr * a.q_rs + h * HD + ch * 8);
          rdo[i] = *reinterpret_cast<const u32x4*>(a.dO + b * a.do_bs + (long)gr * a.do_rs + h * HD + ch * 8);
          ro[i] = *reinterpret_cast<const u32x4*>(a.O + b * a.o_bs + (long)gr * a.o_rs + h * HD + ch * 8);
          rl[i] = (ch == 0 && q < a.Lq) ? a.LSE[((long)b * a.H + h) * a.Lq + q] : 1e30f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int cc = tid + i * 512, row = cc >> 3, ch = cc & 7, q = q0c + row;
          float dl = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) dl += bf_lo(rdo[i][e]) * bf_lo(ro[i][e]) + bf_hi(rdo[i][e]) * bf_hi(ro[i][e]);
          dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64); dl += __shfl_xor(dl, 4, 64);
          const int off = row * 128 + ((ch ^ (row & 7)) << 4);
          *reinterpret_cast<u32x4*>(smem + LP_Q + off) = rq[i];
          *reinterpret_cast<u32x4*>(smem + LP_DO + off) = rdo[i];
          if (ch == 0) {
            const bool ok = q < a.Lq;
            lse_s[row] = rl[i];
            del_s[row] = ok ? dl : 0.f;
            if (ok && part == 0) a.Delta[((long)b * a.H + h) * a.Lq + q] = dl;
          }
        }
      }
      __syncthreads();
      // ---- phase A (attn_bwd_onepass_kernel's, on this part's keys and this chunk's queries)
      auto phase_a = [&](auto ng_) {
        constexpr int NG = decltype(ng_)::value;
        float mk[NG];
#pragma unroll
        for (int gk = 0; gk < NG; ++gk) mk[gk] = (a.mask ? a.mask[(long)b * a.mask_ld + key[gk]] : 0.f) * LOG2E;
        const int last_pair = ((nvs + 1) >> 1) - 1;
        float4 bnext[NG][2];
        auto load_bias = [&](int s) {         // columns < 128 * nchunk <= biasT_ld (checked by the entry point)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) {
              bnext[gk][t] = float4{0.f, 0.f, 0.f, 0.f};
              if (BL2 || a.biasT) bnext[gk][t] = *reinterpret_cast<const float4*>(a.biasT + ((long)h * a.Lk + key[gk]) * a.biasT_ld + q0c + 32 * s + 16 * t + g * 4);
            }
        };
        load_bias(0);
        auto pair = [&](int s, auto full_) {
          constexpr int NTQ = decltype(full_)::value ? 2 : 1;
          float4 btv[NG][2];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) {
              btv[gk][t] = bnext[gk][t];
              if (!BL2 && (a.dbg & 16)) { btv[gk][t].x *= LN2; btv[gk][t].y *= LN2; btv[gk][t].z *= LN2; btv[gk][t].w *= LN2; }
            }
          load_bias(min(s + 1, last_pair));
          f32x4 p[NG][2], ds[NG][2];
#pragma unroll
          for (int gk = 0; gk < NG; ++gk) { p[gk][1] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[gk][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int t = 0; t < NTQ; ++t) {
            f32x4 sa[NG], dp[NG];
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) { sa[gk] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[gk] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const bf16x8 qfr = frag_rows(qtile, 32 * s + 16 * t + fi, ks * 4 + g), dofr = frag_rows(dotile, 32 * s + 16 * t + fi, ks * 4 + g);
#pragma unroll
              for (int gk = 0; gk < NG; ++gk) {
                sa[gk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[gk][ks], sa[gk], 0, 0, 0);
                dp[gk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofr, vf[gk][ks], dp[gk], 0, 0, 0);
              }
            }
            const float4 ls = *reinterpret_cast<const float4*>(lse_s + 32 * s + 16 * t + g * 4);
            const float4 dl = *reinterpret_cast<const float4*>(del_s + 32 * s + 16 * t + g * 4);
            const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) {
              const float bbv[4] = {btv[gk][t].x, btv[gk][t].y, btv[gk][t].z, btv[gk][t].w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float x = BL2 ? fmaf(sa[gk][r], sc2, bbv[r]) : fmaf(sa[gk][r], sc2, fmaf(bbv[r], LOG2E, mk[gk]));
                const float pv = fast_exp2(fminf(x, 1e29f) - lsv[r]);
                p[gk][t][r] = pv;
                ds[gk][t][r] = pv * (dp[gk][r] - dlv[r]);
              }
            }
          }
          bf16x8 pf[NG], dsf[NG];
#pragma unroll
          for (int gk = 0; gk < NG; ++gk) {
            pf[gk] = pack8(p[gk][0], p[gk][1]);
            dsf[gk] = pack8(ds[gk][0], ds[gk][1]);
            const u32x4 w = __builtin_bit_cast(u32x4, dsf[gk]);
            char* blk = smem + LP_DS + (2 * s) * LP_BLK + ((wave + 8 * gk) * 16 + fi) * 32 + g * 8;
            *reinterpret_cast<u32x2*>(blk) = kok[gk] ? u32x2{w[0], w[1]} : u32x2{0u, 0u};
            if (NTQ == 2) *reinterpret_cast<u32x2*>(blk + LP_BLK) = kok[gk] ? u32x2{w[2], w[3]} : u32x2{0u, 0u};
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 dotr = frag_cols(dotile, s, dt, lane), qtr = frag_cols(qtile, s, dt, lane);
#pragma unroll
            for (int gk = 0; gk < NG; ++gk) {
              dv[gk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotr, pf[gk], dv[gk][dt], 0, 0, 0);
              dk[gk][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr, dsf[gk], dk[gk][dt], 0, 0, 0);
            }
          }
        };
        for (int s = 0; s < (nvs >> 1); ++s) pair(s, std::true_type{});
        if (nvs & 1) pair(nvs >> 1, std::false_type{});
      };
      if (ng == 2) phase_a(std::integral_constant<int, 2>{});
      else if (ng == 1) phase_a(std::integral_constant<int, 1>{});
      __syncthreads();
      // ---- phase B: query strip `wave` of the chunk over the part's keys
      if (wave < nvs) {
        const int q = q0c + wave * 16 + fi;
        const int kend = min(nsk * 16, a.ds_ld);
        // the partial dQ of this strip as the earlier parts left it (this very lane wrote it: program order makes it visible), requested
        // now and added behind the products (held through phase A it spilled)
        float4 prev[4];
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
          prev[dt] = float4{0.f, 0.f, 0.f, 0.f};
          if (part > 0) prev[dt] = *reinterpret_cast<const float4*>(wsq + (long)c * (LP_STRIPS * 1024) + dt * 256);
        }
        for (int s = 0; s < ((nkp + 1) >> 1); ++s) {
          const uint32_t a0 = lbase + LP_DS + wave * LP_BLK + (32 * s + 4 * g + r4) * 32 + c4 * 8;
          const bf16x8 dsf = lds_read_tr_frag(a0, a0 + 16 * 32);
          if (a.dS && q < a.Lq) {          // the dS stream of the bias gradient, as attn_bwd_onepass_kernel writes it (columns < min(16 ceil(Lk / 16), ds_ld))
            const u32x4 w = __builtin_bit_cast(u32x4, dsf);
            const int kc = part * LP_KEYS + 32 * s + 4 * g;
            bf16_t* dsp = a.dS + (((long)b * a.H + h) * a.Lq + q) * a.ds_ld + kc;
            if (kc < kend) *reinterpret_cast<u32x2*>(dsp) = u32x2{w[0], w[1]};
            if (kc + 16 < kend) *reinterpret_cast<u32x2*>(dsp + 16) = u32x2{w[2], w[3]};
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(ktile, s, dt, lane), dsf, dq[dt], 0, 0, 0);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dq[dt][0] += prev[dt].x; dq[dt][1] += prev[dt].y; dq[dt][2] += prev[dt].z; dq[dt][3] += prev[dt].w; }
        if (part + 1 < npart) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<float4*>(wsq + (long)c * (LP_STRIPS * 1024) + dt * 256) = float4{dq[dt][0], dq[dt][1], dq[dt][2], dq[dt][3]};
        } else {
          float qs[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) qs[i] = 0.f;
          if (q < a.Lq) {
            bf16_t* op = a.dQ + b * a.dq_bs + (long)q * a.dq_rs + h * HD + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              const u32x2 wq = u32x2{pack_bf16(dq[dt][0] * a.scale, dq[dt][1] * a.scale), pack_bf16(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
              *reinterpret_cast<u32x2*>(op + dt * 16) = wq;
              qs[dt * 4 + 0] = bf_lo(wq[0]); qs[dt * 4 + 1] = bf_hi(wq[0]); qs[dt * 4 + 2] = bf_lo(wq[1]); qs[dt * 4 + 3] = bf_hi(wq[1]);
            }
          }
          if (a.colsum_ws) {           // colsum of the stored dQ: this strip's share onto the wave's slot (chunks in order)
            strip_colsum16(qs);
            if (fi == 0) {
#pragma unroll
              for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) LP_CS_SLOT(0)[dt * 16 + g * 4 + r] += qs[dt * 4 + r];
            }
          }
        }
      }
      __syncthreads();                 // the next chunk overwrites Q / dO / LSE / Delta, its phase A the dS blocks; the next part the K image
    }
    float vs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) vs[i] = 0.f;
#pragma unroll
    for (int gk = 0; gk < 2; ++gk) {
      if (gk >= ng || !kok[gk]) continue;
      bf16_t* kp = a.dK + b * a.dk_bs + (long)key[gk] * a.dk_rs + h * HD + g * 4;
      bf16_t* vp = a.dV + b * a.dv_bs + (long)key[gk] * a.dv_rs + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<u32x2*>(kp + dt * 16) = u32x2{pack_bf16(dk[gk][dt][0] * a.scale, dk[gk][dt][1] * a.scale),
                                                         pack_bf16(dk[gk][dt][2] * a.scale, dk[gk][dt][3] * a.scale)};
        const u32x2 wv = u32x2{pack_bf16(dv[gk][dt][0], dv[gk][dt][1]), pack_bf16(dv[gk][dt][2], dv[gk][dt][3])};
        *reinterpret_cast<u32x2*>(vp + dt * 16) = wv;
        vs[dt * 4 + 0] += bf_lo(wv[0]); vs[dt * 4 + 1] += bf_hi(wv[0]); vs[dt * 4 + 2] += bf_lo(wv[1]); vs[dt * 4 + 3] += bf_hi(wv[1]);
      }
    }
    if (a.colsum_ws) {                 // colsum of the stored dV: this part's share onto the wave's slot (parts in order)
      strip_colsum16(vs);
      if (fi == 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) LP_CS_SLOT(1)[dt * 16 + g * 4 + r] += vs[dt * 4 + r];
      }
    }
  }
  if (a.colsum_ws) {                   // (block-uniform) the eight waves' sums in wave order: see attn_bwd_onepass_kernel
    __syncthreads();
    if (tid < 128) {
      const int kind = tid >> 6, d = tid & 63;
      const float* cs = reinterpret_cast<const float*>(smem + LP_CS) + kind * 8 * 64 + d;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += cs[w * 64];
      a.colsum_ws[((long)b * 2 + kind) * (a.H * HD) + h * HD + d] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------ C ABI
// `args` is the AttnArgs struct laid out as 8-byte slots (pointers, longs) followed by ints/floats;
// the Python side fills it through ctypes.Structure with the same field order.
// Kernel selection is fixed (round 4 pruned the run-time variant bits; what they measured is in DESIGN section 5.1 / 5.3): strip-walking
// resident forward / dQ kernels at Lk <= WALK_ROWS, two-workgroup resident kernels up to 256 keys, 8-wave streaming kernels
// beyond, per-row forward + grouped dQ for rows that share K/V, XCD-aware block order for the kernels that read a bias.
// Launch `kernel` over the logical grid (nx tiles, ny heads, nz batches): as a 3-D grid, or - xmap, the kernels that read a
// relative-position bias - as the 1-D XCD-aware grid attn_block() decodes.
template <typename Kern>
static void attn_launch(Kern kernel, AttnArgs a, int nx, int ny, int nz, int threads, bool xmap, hipStream_t st) {
  a.grid_nx = nx; a.grid_ny = ny; a.grid_nz = nz; a.grid_map = 0;
  if (xmap) {
    int C = 1;
    while (C < 8 && (ny * C) % 8 != 0) C *= 2;        // batch chunks per head: (head, chunk) units divide evenly over 8 XCDs
    if (C > nz) C = 1;
    a.grid_map = C;
    const int zc = (nz + C - 1) / C, units = ny * C;
    hipLaunchKernelGGL(kernel, dim3(8 * ((units + 7) / 8) * zc * nx), dim3(threads), 0, st, a);
    return;
  }
  hipLaunchKernelGGL(kernel, dim3(nx, ny, nz), dim3(threads), 0, st, a);
}

static int check_common(const AttnArgs& a, const char* who) {
  X2_REQUIRE(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0, "%s: empty problem", who);
  X2_REQUIRE(a.head_dim == HD, "%s: head_dim %d is not supported (kernels are built for %d)", who, a.head_dim, HD);
  X2_REQUIRE(!a.bias || (a.bias_ld % 64 == 0 && a.bias_ld >= a.Lk), "%s: bias_ld must be a multiple of 64 covering Lk", who);
  X2_REQUIRE(!a.mask || (a.mask_ld % 64 == 0 && a.mask_ld >= a.Lk), "%s: mask_ld must be a multiple of 64 covering Lk", who);
  X2_REQUIRE((a.q_rs % 8 | a.k_rs % 8 | a.v_rs % 8 | a.q_bs % 8 | a.k_bs % 8 | a.v_bs % 8) == 0, "%s: strides must keep 16-byte rows", who);
  return X2_OK;
}

extern "C" int x2_attn_fwd(const AttnArgs* pa, void* stream) {
  AttnArgs a = *pa;
  a.grid_nx = a.grid_ny = a.grid_nz = a.grid_map = 0;
  if (int e = check_common(a, "x2_attn_fwd")) return e;
  const bool xm = a.bias != nullptr && !a.kv_idx;        // XCD-aware block order for the kernels that read a [H][Lq][Lk] bias
  const bool bl2 = (a.dbg & 16) && a.bias && !a.mask;   // bias handed over in log2 units (kernels.relpos_bias(log2=True)), no mask
  X2_REQUIRE(a.Q && a.K && a.V && a.Out && a.LSE, "x2_attn_fwd: null tensor");
  X2_REQUIRE((a.o_rs % 4 | a.o_bs % 4) == 0, "x2_attn_fwd: output strides");
  const hipStream_t st = (hipStream_t)stream;
  // long query side and <= 4 key tiles: 8-wave workgroups with K/V resident in LDS (64 KB); otherwise key tiles are
  // streamed through a double buffer (short query side: the 64 KB would leave 2 waves / workgroup alone on a CU)
  X2_REQUIRE(!a.seq_off || (a.seq_ids && a.Bkv > 0), "x2_attn_fwd: seq_off needs seq_ids and Bkv");
  if (a.Lq <= 32 && a.Lk <= 64) hipLaunchKernelGGL((attn_fwd_kernel<2, 1, true, 1>), dim3(1, a.H, a.B), dim3(128), 0, st, a);
  else if (a.Lq <= 32) hipLaunchKernelGGL((attn_fwd_kernel<2, 1, false>), dim3(1, a.H, a.B), dim3(128), 0, st, a);
  else if (a.Lk > 256 && a.Lq > 64)      // long sequences (X2VLM-large, N = 577): 8 waves share each streamed K/V tile (285 -> 254 us)
    // (two query strips per wave sharing the K / V fragment reads - <4, 2> and <8, 2> at two waves per SIMD - measured 176-190 and 215-231 us against 160-170, round 6)
    { if (bl2) attn_launch(attn_fwd_kernel<8, 1, false, 2, 4, true>, a, (a.Lq + 127) / 128, a.H, a.B, 512, xm, st);
      else attn_launch(attn_fwd_kernel<8, 1, false>, a, (a.Lq + 127) / 128, a.H, a.B, 512, xm, st); }
  else if (a.Lq <= 64 || a.Lk > 256) hipLaunchKernelGGL((attn_fwd_kernel<4, 1, false>), dim3((a.Lq + 63) / 64, a.H, a.B), dim3(256), 0, st, a);
  else if (!a.kv_idx && a.Lk <= WALK_ROWS)      // staged: one strip-walking workgroup per (sequence, head)
    { if (bl2) attn_launch(attn_fwd_walk_kernel<4, true>, a, 1, a.H, a.B, 256, xm, st);
      else attn_launch(attn_fwd_walk_kernel<4>, a, 1, a.H, a.B, 256, xm, st); }
  else attn_launch(attn_fwd_kernel<8, 1, true>, a, (a.Lq + 127) / 128, a.H, a.B, 512, xm, st);
  return x2_check_launch("x2_attn_fwd");
}

// Which backward runs for these arguments (3: see attn_bwd_one_pass_long below): 0 = the dQ + dK/dV pair, 1 = attn_bwd_onepass_kernel (one sequence per K/V batch, 64 < Lq,
// Lk <= 208, no probability dropout), 2 = attn_bwd_onepass_grouped_kernel (rows sharing K/V through the CSR, Lq <= 128, Lk <= 208, no
// bias).  Only a phase-0 call can take the one-pass forms; x2_tune(14, 1) switches them off (A/B measurements, tests of both forms).
static int attn_bwd_one_pass(const AttnArgs& a) {
  if (a.phase != 0 || x2_tune_get(14) == 1 || a.Lk > OP_ROWS) return 0;
  if ((a.do_rs % 8 | a.do_bs % 8 | a.o_rs % 8 | a.o_bs % 8) != 0) return 0;
  if (!a.seq_off && !a.kv_idx && a.B == a.Bkv && a.Lq > 64 && a.Lq <= OP_ROWS && a.Lk > 64 && !a.drop.thr16) return 1;
  if (a.seq_off && a.seq_ids && a.kv_idx && a.Lq <= XP_ROWS && !a.bias && !a.dS) return 2;
  return 0;
}
// 3 = attn_bwd_onepass_long_kernel (one sequence per K/V batch, 208 < Lq <= 640, 208 < Lk <= 768, no probability dropout, a workspace for
// the dQ partials, the transposed bias padded to whole 128-query chunks)
static int attn_bwd_one_pass_long(const AttnArgs& a) {
  if (a.phase != 0 || x2_tune_get(14) == 1 || x2_tune_get(14) == 3 || a.Lk <= OP_ROWS || a.Lq <= OP_ROWS || a.Lk > LP_MAX_LK || a.Lq > LP_MAX_LQ) return 0;
  if ((a.do_rs % 8 | a.do_bs % 8 | a.o_rs % 8 | a.o_bs % 8) != 0) return 0;
  if (a.seq_off || a.kv_idx || a.B != a.Bkv || a.drop.thr16) return 0;
  const long chunks = (a.Lq + LP_ROWS - 1) / LP_ROWS;
  if (!a.ws || a.ws_floats < (long)a.B * a.H * chunks * LP_STRIPS * 1024) return 0;
  if (a.biasT && a.biasT_ld < chunks * LP_ROWS) return 0;
  return 3;
}
// for callers that place the K/V-side half on another stream only when there is one (engine.py): 1 / 2 as above when a phase-0 call
// with these arguments runs in one pass
extern "C" int x2_attn_bwd_one_pass(const AttnArgs* pa) {
  AttnArgs a = *pa; a.phase = 0;
  if (const int f = attn_bwd_one_pass(a)) return f;
  return attn_bwd_one_pass_long(a);
}

extern "C" int x2_attn_bwd(const AttnArgs* pa, void* stream) {
  AttnArgs a = *pa;
  a.grid_nx = a.grid_ny = a.grid_nz = a.grid_map = 0;
  if (int e = check_common(a, "x2_attn_bwd")) return e;
  const bool xm = a.bias != nullptr && !a.kv_idx, xmT = a.biasT != nullptr && !a.seq_off;
  const bool bl2 = (a.dbg & 16) && a.bias && a.biasT && !a.mask;
  X2_REQUIRE(a.Q && a.K && a.V && a.O && a.dO && a.dQ && a.dK && a.dV && a.LSE && a.Delta, "x2_attn_bwd: null tensor");
  X2_REQUIRE(!a.bias || (a.biasT && a.biasT_ld % 64 == 0 && a.biasT_ld >= a.Lq), "x2_attn_bwd: biasT [H][Lk][ld%%64==0] required with bias");
  X2_REQUIRE(!a.dS || (a.ds_ld % 64 == 0 && a.ds_ld >= a.Lk), "x2_attn_bwd: ds_ld must be a multiple of 64 covering Lk");
  X2_REQUIRE((a.kv_idx == nullptr) == (a.seq_off == nullptr), "x2_attn_bwd: kv_idx and seq_off/seq_ids come together");
  X2_REQUIRE(a.Bkv > 0, "x2_attn_bwd: Bkv");
  X2_REQUIRE(a.phase >= 0 && a.phase <= 2, "x2_attn_bwd: phase=%d", a.phase);
  const hipStream_t st = (hipStream_t)stream;
  // one workgroup per (sequence, head) for the whole backward (attn_bwd_onepass_kernel): the BEiT-2 blocks at N = 197; x2_tune(14, 1)
  // keeps the dQ + dK/dV pair (A/B measurements, and the tests that compare the two forms)
  if (attn_bwd_one_pass(a) == 2) {
    if (a.drop.thr16) hipLaunchKernelGGL((attn_bwd_onepass_grouped_kernel<true>), dim3(1, a.H, a.Bkv), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((attn_bwd_onepass_grouped_kernel<false>), dim3(1, a.H, a.Bkv), dim3(512), 0, st, a);
    return x2_check_launch("x2_attn_bwd(one pass, shared K/V)");
  }
  if (attn_bwd_one_pass(a) == 1) {
    if (bl2) attn_launch(attn_bwd_onepass_kernel<true>, a, 1, a.H, a.B, 512, xm, st);
    else attn_launch(attn_bwd_onepass_kernel<false>, a, 1, a.H, a.B, 512, xm, st);
    return x2_check_launch("x2_attn_bwd(one pass)");
  }
  if (attn_bwd_one_pass_long(a) == 3) {
    if (bl2) attn_launch(attn_bwd_onepass_long_kernel<true>, a, 1, a.H, a.B, 512, xm, st);
    else attn_launch(attn_bwd_onepass_long_kernel<false>, a, 1, a.H, a.B, 512, xm, st);
    return x2_check_launch("x2_attn_bwd(one pass, long)");
  }
  if (a.phase == 2) { /* the dQ half ran in an earlier call */ }
  else if (a.seq_off && a.Lk <= 256 && !a.bias && !a.dS)      // rows sharing K/V: one workgroup per (K/V batch, head)
    hipLaunchKernelGGL((attn_bwd_dq_grouped_kernel<8>), dim3(1, a.H, a.Bkv), dim3(512), 0, st, a);
  // (the 30-token self-attentions stay two launches: dQ and dK / dV fused into one two-wave workgroup per (sequence, head) - Q / dO / K staged once,
  // no Delta round trip - measured 22.80 vs 22.70 ms per base step, profiles/r12e_attn_short_fused_ab.txt: these launches are a load -> multiply ->
  // store chain per workgroup, and one workgroup doing both halves is that chain twice as long on half as many workgroups.  Removed.)
  else if (a.Lq <= 32 && a.Lk <= 64) hipLaunchKernelGGL((attn_bwd_dq_kernel<2, 1, true, 1>), dim3(1, a.H, a.B), dim3(128), 0, st, a);
  else if (a.Lq <= 32) hipLaunchKernelGGL((attn_bwd_dq_kernel<2, 1, false>), dim3(1, a.H, a.B), dim3(128), 0, st, a);
  else if (a.Lk > 256 && a.Lq > 64)      // N = 577: dQ + dK/dV 810 -> 652 us with 8-wave workgroups
    { if (bl2) attn_launch(attn_bwd_dq_kernel<8, 1, false, 2, 4, true>, a, (a.Lq + 127) / 128, a.H, a.B, 512, xm, st);
      else attn_launch(attn_bwd_dq_kernel<8, 1, false>, a, (a.Lq + 127) / 128, a.H, a.B, 512, xm, st); }
  else if (a.Lq <= 64 || a.Lk > 256) hipLaunchKernelGGL((attn_bwd_dq_kernel<4, 1, false>), dim3((a.Lq + 63) / 64, a.H, a.B), dim3(256), 0, st, a);
  else if (!a.kv_idx && a.Lk <= WALK_ROWS)
    { if (bl2) attn_launch(attn_bwd_dq_walk_kernel<4, true>, a, 1, a.H, a.B, 256, xm, st);
      else attn_launch(attn_bwd_dq_walk_kernel<4>, a, 1, a.H, a.B, 256, xm, st); }
  else attn_launch(attn_bwd_dq_kernel<8, 1, true>, a, (a.Lq + 127) / 128, a.H, a.B, 512, xm, st);
  if (int e = x2_check_launch("x2_attn_bwd(dq)")) return e;
  if (a.phase == 1) return 0;
  const bool res = !a.seq_off && a.Lq > 64 && a.Lq <= 256;   // one sequence per K/V batch, 2..4 query tiles: resident Q/dO
  if (a.Lk <= 32 && !a.seq_off && a.Lq <= 64) {          // text self-attention: one sequence, one query tile, 17 KB of LDS
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, 1, true, 1>), dim3(1, a.H, a.Bkv), dim3(128), 0, st, a);
  } else if (a.Lk <= 32) {
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, 1, false>), dim3(1, a.H, a.Bkv), dim3(128), 0, st, a);
  } else if (res) {
    // 8 waves (128 keys) share the resident Q / dO image (133 KB: one workgroup per CU either way): 206 -> 189 us per
    // vision layer against 4-wave workgroups, which left 4 waves on a CU
    if (bl2) attn_launch(attn_bwd_dkv_kernel<8, 1, true, 4, 4, true>, a, (a.Lk + 127) / 128, a.H, a.Bkv, 512, xmT, st);
    else attn_launch(attn_bwd_dkv_kernel<8, 1, true>, a, (a.Lk + 127) / 128, a.H, a.Bkv, 512, xmT, st);
  } else if (a.Lk > 256 && !a.seq_off) {
    if (bl2) attn_launch(attn_bwd_dkv_kernel<8, 1, false, 2, 4, true>, a, (a.Lk + 127) / 128, a.H, a.Bkv, 512, xmT, st);
    else attn_launch(attn_bwd_dkv_kernel<8, 1, false>, a, (a.Lk + 127) / 128, a.H, a.Bkv, 512, xmT, st);
  } else {
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<4, 1, false>), dim3((a.Lk + 63) / 64, a.H, a.Bkv), dim3(256), 0, st, a);
  }
  return x2_check_launch("x2_attn_bwd(dkv)");
}
