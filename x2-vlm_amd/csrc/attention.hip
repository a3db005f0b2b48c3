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
// reduced from) and dK/dV.
#include "x2_common.h"

#define HD 64                 // head dim
#define KT 64                 // keys (or queries) per LDS tile
#define NEG_BIG (-1.0e30f)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

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
};

// stage a [64 rows][64 d] bf16 tile (rows clamped to `nrows-1`) into LDS, chunk c of row r at c ^ (r & 7)
template <int NT>
__device__ __forceinline__ void load_tile(char* lds, const bf16_t* src, long rs, int row0, int nrows, int tid) {
#pragma unroll
  for (int c = tid; c < 512; c += NT) {
    const int r = c >> 3, ch = c & 7;
    int gr = row0 + r; gr = gr < nrows ? gr : nrows - 1;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + (long)gr * rs + ch * 8);
    *reinterpret_cast<u32x4*>(lds + r * 128 + ((ch ^ (r & 7)) << 4)) = v;
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
  if (a.bias) bb = *reinterpret_cast<const float4*>(a.bias + ((long)h * a.Lq + q) * a.bias_ld + key0);
  if (a.mask) mm = *reinterpret_cast<const float4*>(a.mask + (long)b * a.mask_ld + key0);
  f32x4 o;
  o[0] = key0 + 0 < a.Lk ? s[0] * sc2 + (bb.x + mm.x) * LOG2E : NEG_BIG;
  o[1] = key0 + 1 < a.Lk ? s[1] * sc2 + (bb.y + mm.y) * LOG2E : NEG_BIG;
  o[2] = key0 + 2 < a.Lk ? s[2] * sc2 + (bb.z + mm.z) * LOG2E : NEG_BIG;
  o[3] = key0 + 3 < a.Lk ? s[3] * sc2 + (bb.w + mm.w) * LOG2E : NEG_BIG;
  return o;
}

// ------------------------------------------------------------------------------------------ forward
template <int QW>
__global__ __launch_bounds__(64 * QW) void attn_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * KT * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, bk = a.kv_idx ? a.kv_idx[b] : b;
  const int q0 = blockIdx.x * 16 * QW + wave * 16;
  const int q = min(q0 + fi, a.Lq - 1);
  const uint32_t ktile = lds_addr(smem), vtile = ktile + KT * 128;
  const bf16_t* Kp = a.K + bk * a.k_bs + h * HD;
  const bf16_t* Vp = a.V + bk * a.v_bs + h * HD;
  const float sc2 = a.scale * LOG2E;

  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(a.Q + b * a.q_bs + (long)q * a.q_rs + h * HD + ks * 32 + g * 8);
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_i = NEG_BIG, l_i = 0.f;

  const int nkt = (a.Lk + KT - 1) / KT;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    load_tile<64 * QW>(smem, Kp, a.k_rs, kt * KT, a.Lk, tid);
    load_tile<64 * QW>(smem + KT * 128, Vp, a.v_rs, kt * KT, a.Lk, tid);
    __syncthreads();
    f32x4 st[4];
    float mx = NEG_BIG;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(ktile, nt * 16 + fi, ks * 4 + g), qf[ks], acc, 0, 0, 0);
      st[nt] = add_bias_mask(acc, a, h, b, q, kt * KT + nt * 16 + g * 4, sc2);
      mx = fmaxf(fmaxf(mx, fmaxf(st[nt][0], st[nt][1])), fmaxf(st[nt][2], st[nt][3]));
    }
    mx = group_max(mx);
    const float m_new = fmaxf(m_i, mx);
    const float alpha = exp2f(m_i - m_new);
    float rs = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[nt][r] = exp2f(st[nt][r] - m_new); rs += st[nt][r]; }
    l_i = l_i * alpha + group_sum(rs);
    m_i = m_new;
    if (a.drop.thr16) {      // normalisation uses the undropped sum; only the P that multiplies V is dropped
      const uint32_t e0 = (uint32_t)(((long)b * a.H + h) * a.Lq + q) * (uint32_t)((a.Lk + 63) & ~63) + (uint32_t)(kt * KT + g * 4);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float dm[4];
        drop_mul4(a.drop, e0 + nt * 16, dm);
        st[nt][0] *= dm[0]; st[nt][1] *= dm[1]; st[nt][2] *= dm[2]; st[nt][3] *= dm[3];
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 pf = pack8(st[2 * s], st[2 * s + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(vtile, s, dt, lane), pf, o[dt], 0, 0, 0);
    }
  }
  if (q0 + fi < a.Lq) {
    const float inv = 1.0f / l_i;
    bf16_t* op = a.Out + b * a.o_bs + (long)q * a.o_rs + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<u32x2*>(op + dt * 16) = u32x2{pack_bf16(o[dt][0] * inv, o[dt][1] * inv), pack_bf16(o[dt][2] * inv, o[dt][3] * inv)};
    if (g == 0) a.LSE[((long)b * a.H + h) * a.Lq + q] = m_i + log2f(l_i);   // log2 domain
  }
}

// ------------------------------------------------------------------------------------------ backward: dQ (+ dS)
template <int QW>
__global__ __launch_bounds__(64 * QW) void attn_bwd_dq_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * KT * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z, bk = a.kv_idx ? a.kv_idx[b] : b;
  const int q0 = blockIdx.x * 16 * QW + wave * 16;
  const int q = min(q0 + fi, a.Lq - 1);
  const bool qvalid = q0 + fi < a.Lq;
  const uint32_t ktile = lds_addr(smem), vtile = ktile + KT * 128;
  const bf16_t* Kp = a.K + bk * a.k_bs + h * HD;
  const bf16_t* Vp = a.V + bk * a.v_bs + h * HD;
  const float sc2 = a.scale * LOG2E;

  bf16x8 qf[2], dof[2];
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(a.Q + b * a.q_bs + (long)q * a.q_rs + h * HD + ks * 32 + g * 8);
    dof[ks] = *reinterpret_cast<const bf16x8*>(a.dO + b * a.do_bs + (long)q * a.do_rs + h * HD + ks * 32 + g * 8);
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(a.O + b * a.o_bs + (long)q * a.o_rs + h * HD + ks * 32 + g * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) delta += bf2f((bf16_t)dof[ks][e]) * bf2f((bf16_t)of[e]);
  }
  delta = group_sum(delta);
  const float lse = a.LSE[((long)b * a.H + h) * a.Lq + q];
  if (qvalid && g == 0) a.Delta[((long)b * a.H + h) * a.Lq + q] = delta;

  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (a.Lk + KT - 1) / KT;
  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();
    load_tile<64 * QW>(smem, Kp, a.k_rs, kt * KT, a.Lk, tid);
    load_tile<64 * QW>(smem + KT * 128, Vp, a.v_rs, kt * KT, a.Lk, tid);
    __syncthreads();
    f32x4 ds[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 s{0.f, 0.f, 0.f, 0.f}, dp{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(ktile, nt * 16 + fi, ks * 4 + g), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(vtile, nt * 16 + fi, ks * 4 + g), dof[ks], dp, 0, 0, 0);
      }
      const int key0 = kt * KT + nt * 16 + g * 4;
      s = add_bias_mask(s, a, h, b, q, key0, sc2);
      if (a.drop.thr16) {
        float dm[4];
        drop_mul4(a.drop, (uint32_t)(((long)b * a.H + h) * a.Lq + q) * (uint32_t)((a.Lk + 63) & ~63) + (uint32_t)key0, dm);
        dp[0] *= dm[0]; dp[1] *= dm[1]; dp[2] *= dm[2]; dp[3] *= dm[3];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[nt][r] = exp2f(s[r] - lse) * (dp[r] - delta);
      if (a.dS && qvalid && key0 < a.ds_ld)
        *reinterpret_cast<u32x2*>(a.dS + (((long)b * a.H + h) * a.Lq + q) * a.ds_ld + key0) =
            u32x2{pack_bf16(ds[nt][0], ds[nt][1]), pack_bf16(ds[nt][2], ds[nt][3])};
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const bf16x8 dsf = pack8(ds[2 * s2], ds[2 * s2 + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(ktile, s2, dt, lane), dsf, dq[dt], 0, 0, 0);
    }
  }
  if (qvalid) {
    bf16_t* op = a.dQ + b * a.dq_bs + (long)q * a.dq_rs + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<u32x2*>(op + dt * 16) = u32x2{pack_bf16(dq[dt][0] * a.scale, dq[dt][1] * a.scale),
                                                       pack_bf16(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
  }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
template <int KW>
__global__ __launch_bounds__(64 * KW) void attn_bwd_dkv_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * KT * 128 + 2 * KT * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fi = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, bk = blockIdx.z;
  const int k0 = blockIdx.x * 16 * KW + wave * 16;
  const int key = min(k0 + fi, a.Lk - 1);
  const bool kvalid = k0 + fi < a.Lk;
  const uint32_t qtile = lds_addr(smem), dotile = qtile + KT * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * KT * 128);
  float* del_s = lse_s + KT;
  const float sc2 = a.scale * LOG2E;

  bf16x8 kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(a.K + bk * a.k_bs + (long)key * a.k_rs + h * HD + ks * 32 + g * 8);
    vf[ks] = *reinterpret_cast<const bf16x8*>(a.V + bk * a.v_bs + (long)key * a.v_rs + h * HD + ks * 32 + g * 8);
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int sb = a.seq_off ? a.seq_off[bk] : bk, se = a.seq_off ? a.seq_off[bk + 1] : bk + 1;
  const int nqt = (a.Lq + KT - 1) / KT;
  for (int si = sb; si < se; ++si) {
    const int b = a.seq_ids ? a.seq_ids[si] : si;
    const float mk = (a.mask ? a.mask[(long)b * a.mask_ld + key] : 0.f) * LOG2E;
    for (int qt = 0; qt < nqt; ++qt) {
      __syncthreads();
      load_tile<64 * KW>(smem, a.Q + b * a.q_bs + h * HD, a.q_rs, qt * KT, a.Lq, tid);
      load_tile<64 * KW>(smem + KT * 128, a.dO + b * a.do_bs + h * HD, a.do_rs, qt * KT, a.Lq, tid);
      if (tid < KT) {
        const int qq = min(qt * KT + tid, a.Lq - 1);
        lse_s[tid] = a.LSE[((long)b * a.H + h) * a.Lq + qq];
        del_s[tid] = a.Delta[((long)b * a.H + h) * a.Lq + qq];
      }
      __syncthreads();
      f32x4 p[4], ds[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 s{0.f, 0.f, 0.f, 0.f}, dp{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(qtile, t * 16 + fi, ks * 4 + g), kf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(dotile, t * 16 + fi, ks * 4 + g), vf[ks], dp, 0, 0, 0);
        }
        // this lane: key = its own, queries qq0 .. qq0+3
        const int qq0 = qt * KT + t * 16 + g * 4;
        float4 bb{0.f, 0.f, 0.f, 0.f};
        if (a.biasT) bb = *reinterpret_cast<const float4*>(a.biasT + ((long)h * a.Lk + key) * a.biasT_ld + qq0);
        const float4 ls = *reinterpret_cast<const float4*>(lse_s + t * 16 + g * 4);
        const float4 dl = *reinterpret_cast<const float4*>(del_s + t * 16 + g * 4);
        const float bbv[4] = {bb.x, bb.y, bb.z, bb.w}, lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = kvalid && (qq0 + r < a.Lq);
          const float pv = ok ? exp2f(s[r] * sc2 + bbv[r] * LOG2E + mk - lsv[r]) : 0.f;
          float dm = 1.f;
          if (a.drop.thr16)
            dm = drop_mul(a.drop, (uint32_t)(((long)b * a.H + h) * a.Lq + min(qq0 + r, a.Lq - 1)) * (uint32_t)((a.Lk + 63) & ~63) + (uint32_t)key);
          p[t][r] = pv * dm;
          ds[t][r] = pv * (dp[r] * dm - dlv[r]);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 pf = pack8(p[2 * s2], p[2 * s2 + 1]);
        const bf16x8 dsf = pack8(ds[2 * s2], ds[2 * s2 + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(dotile, s2, dt, lane), pf, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(qtile, s2, dt, lane), dsf, dk[dt], 0, 0, 0);
        }
      }
    }
  }
  if (kvalid) {
    bf16_t* kp = a.dK + bk * a.dk_bs + (long)key * a.dk_rs + h * HD + g * 4;
    bf16_t* vp = a.dV + bk * a.dv_bs + (long)key * a.dv_rs + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<u32x2*>(kp + dt * 16) = u32x2{pack_bf16(dk[dt][0] * a.scale, dk[dt][1] * a.scale),
                                                       pack_bf16(dk[dt][2] * a.scale, dk[dt][3] * a.scale)};
      *reinterpret_cast<u32x2*>(vp + dt * 16) = u32x2{pack_bf16(dv[dt][0], dv[dt][1]), pack_bf16(dv[dt][2], dv[dt][3])};
    }
  }
}

// ------------------------------------------------------------------------------------------ C ABI
// `args` is the AttnArgs struct laid out as 8-byte slots (pointers, longs) followed by ints/floats;
// the Python side fills it through ctypes.Structure with the same field order.
static int check_common(const AttnArgs& a, const char* who) {
  X2_REQUIRE(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0, "%s: empty problem", who);
  X2_REQUIRE(!a.bias || (a.bias_ld % 64 == 0 && a.bias_ld >= a.Lk), "%s: bias_ld must be a multiple of 64 covering Lk", who);
  X2_REQUIRE(!a.mask || (a.mask_ld % 64 == 0 && a.mask_ld >= a.Lk), "%s: mask_ld must be a multiple of 64 covering Lk", who);
  X2_REQUIRE((a.q_rs % 8 | a.k_rs % 8 | a.v_rs % 8 | a.q_bs % 8 | a.k_bs % 8 | a.v_bs % 8) == 0, "%s: strides must keep 16-byte rows", who);
  return X2_OK;
}

extern "C" int x2_attn_fwd(const AttnArgs* pa, void* stream) {
  const AttnArgs a = *pa;
  if (int e = check_common(a, "x2_attn_fwd")) return e;
  X2_REQUIRE(a.Q && a.K && a.V && a.Out && a.LSE, "x2_attn_fwd: null tensor");
  X2_REQUIRE((a.o_rs % 4 | a.o_bs % 4) == 0, "x2_attn_fwd: output strides");
  if (a.Lq <= 32) hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3((a.Lq + 31) / 32, a.H, a.B), dim3(128), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3((a.Lq + 63) / 64, a.H, a.B), dim3(256), 0, (hipStream_t)stream, a);
  return x2_check_launch("x2_attn_fwd");
}

extern "C" int x2_attn_bwd(const AttnArgs* pa, void* stream) {
  const AttnArgs a = *pa;
  if (int e = check_common(a, "x2_attn_bwd")) return e;
  X2_REQUIRE(a.Q && a.K && a.V && a.O && a.dO && a.dQ && a.dK && a.dV && a.LSE && a.Delta, "x2_attn_bwd: null tensor");
  X2_REQUIRE(!a.bias || (a.biasT && a.biasT_ld % 64 == 0 && a.biasT_ld >= a.Lq), "x2_attn_bwd: biasT [H][Lk][ld%%64==0] required with bias");
  X2_REQUIRE(!a.dS || (a.ds_ld % 64 == 0 && a.ds_ld >= a.Lk), "x2_attn_bwd: ds_ld must be a multiple of 64 covering Lk");
  X2_REQUIRE((a.kv_idx == nullptr) == (a.seq_off == nullptr), "x2_attn_bwd: kv_idx and seq_off/seq_ids come together");
  X2_REQUIRE(a.Bkv > 0, "x2_attn_bwd: Bkv");
  if (a.Lq <= 32) hipLaunchKernelGGL(attn_bwd_dq_kernel<2>, dim3((a.Lq + 31) / 32, a.H, a.B), dim3(128), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel<4>, dim3((a.Lq + 63) / 64, a.H, a.B), dim3(256), 0, (hipStream_t)stream, a);
  if (int e = x2_check_launch("x2_attn_bwd(dq)")) return e;
  if (a.Lk <= 32) hipLaunchKernelGGL(attn_bwd_dkv_kernel<2>, dim3((a.Lk + 31) / 32, a.H, a.Bkv), dim3(128), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(attn_bwd_dkv_kernel<4>, dim3((a.Lk + 63) / 64, a.H, a.Bkv), dim3(256), 0, (hipStream_t)stream, a);
  return x2_check_launch("x2_attn_bwd(dkv)");
}
