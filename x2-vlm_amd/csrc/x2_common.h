// Shared device/host helpers for the X^2-VLM gfx950 kernels.  gfx950 (CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                               // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;              // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;               // one 16x16 accumulator tile
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define X2_OK 0
#define X2_ERR_ARG (-1)
#define X2_ERR_LAUNCH (-2)

void x2_set_error(const char* fmt, ...);
int x2_check_launch(const char* what);

#define X2_REQUIRE(cond, ...) do { if (!(cond)) { x2_set_error(__VA_ARGS__); return X2_ERR_ARG; } } while (0)

// ---- bf16 <-> f32 (round to nearest even) ----
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
// two f32 -> packed bf16x2 (lo = a, hi = b), hardware RNE convert (probes/probe_isa.hip: CVTPK)
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// erf-form GELU and its derivative in fp32.  Phi(-|x|) = erfc(|x| / sqrt 2) / 2 by Abramowitz-Stegun 7.1.26 (|error| <= 0.75e-7,
// below fp32 resolution of the surrounding arithmetic) on one v_exp + one v_rcp: ~15 VALU instructions per element instead of
// libm erff's ~50 - these run in the epilogues of the fc1 / GELU' GEMMs, 38.7 M elements per launch, where every VALU
// instruction per element is ~1 us of a ~90 us launch.  (v_rcp_f32 directly: `1.0f / x` and __frcp_rn are the IEEE division,
// ten instructions; the 0.5 and the 1 / sqrt 2 are folded into the constants; the tail form x*Phi(x) = max(x, 0) - |x| Phi(-|x|)
// needs no sign transfer and has no 1 - (1 - small) cancellation for negative x.)  The same exponential exp(-x^2 / 2) serves
// the Gaussian term of the derivative.
__device__ __forceinline__ float gelu_tail(float ax, float e) {            // Phi(-ax), ax = |x|, e = exp(-ax * ax / 2)
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.0f));
  const float poly = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f))));
  return poly * e;
}
__device__ __forceinline__ float gelu_f(float x) {
  const float ax = fabsf(x);
  const float h = gelu_tail(ax, __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f));
  return fmaf(-ax, h, fmaxf(x, 0.0f));
}
__device__ __forceinline__ float dgelu_f(float x) {
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);
  const float h = gelu_tail(fabsf(x), e);
  return fmaf(x * 0.3989422804014327f, e, x >= 0.0f ? 1.0f - h : h);
}

// ---- counter-based dropout: keep(element) is a pure function of (site seed, element index), so the backward
// regenerates the forward's mask instead of storing it.  One 32-bit hash (lowbias32) yields two 16-bit uniforms:
// elements 2k and 2k+1 share a hash.  Dropped when u16 < thr16 (thr16 = round(p * 65536)); thr16 == 0: off.
// Mirrored bit-for-bit by kernels.dropout_keep() on the host for the parity tests.
struct DropSpec { uint32_t thr16; uint32_t seed; float scale; };
__device__ __forceinline__ uint32_t x2_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// Epoch: a step counter in DEVICE memory mixed into the site seed.  Lets a hipGraph-captured step (whose kernel
// arguments are frozen at capture) draw fresh masks on every replay: the graph increments the counter, every dropout
// site of the step reads it.  epoch == NULL: the seed is used as given (eager launches with host-drawn seeds).
// Mirrored by kernels.dropout_keep() on the host.
__device__ __forceinline__ DropSpec drop_at_epoch(DropSpec d, const uint32_t* __restrict__ epoch) {
  if (epoch) d.seed = x2_hash(d.seed + 0x9E3779B1u * epoch[0]);
  return d;
}
__device__ __forceinline__ float drop_mul(const DropSpec& d, uint32_t e) {          // one element
  const uint32_t h = x2_hash((e >> 1) ^ d.seed);
  const uint32_t u = (e & 1u) ? (h >> 16) : (h & 0xffffu);
  return u >= d.thr16 ? d.scale : 0.f;
}
// four consecutive elements e0..e0+3 (e0 % 4 == 0): two hashes
__device__ __forceinline__ void drop_mul4(const DropSpec& d, uint32_t e0, float m[4]) {
  const uint32_t h0 = x2_hash((e0 >> 1) ^ d.seed), h1 = x2_hash(((e0 >> 1) + 1u) ^ d.seed);
  m[0] = (h0 & 0xffffu) >= d.thr16 ? d.scale : 0.f;
  m[1] = (h0 >> 16) >= d.thr16 ? d.scale : 0.f;
  m[2] = (h1 & 0xffffu) >= d.thr16 ? d.scale : 0.f;
  m[3] = (h1 >> 16) >= d.thr16 ? d.scale : 0.f;
}
// ---- wave64 reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- LDS helpers ----
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }

// 16-byte read of an MFMA fragment
__device__ __forceinline__ bf16x8 lds_read_b128(uint32_t byte_addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((uintptr_t)byte_addr);
}
// Transposing read (gfx950): within each 16-lane group, lane 4r+c supplies the address of 4 bf16 at
// [row r][cols 4c..4c+3]; lane i receives {[0][i],[1][i],[2][i],[3][i]}.  Verified by probes/probe_isa.hip.
__device__ __forceinline__ bf16x4 lds_read_tr64(uint32_t byte_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4*)(uintptr_t)byte_addr);
}
// two transposing reads -> one 8-element MFMA fragment (slots 0..3 from `a0`, 4..7 from `a1`)
__device__ __forceinline__ bf16x8 lds_read_tr_frag(uint32_t a0, uint32_t a1) {
  const bf16x4 lo = lds_read_tr64(a0), hi = lds_read_tr64(a1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// The same fragment read as inline asm.  hipcc treats the ds_read_tr builtin as touching unknown memory and puts an
// s_waitcnt vmcnt(0) in front of it whenever a global->LDS copy is in flight, i.e. it drains the prefetch of the next
// tile before every fragment read.  The asm form is invisible to that analysis: the CALLER orders it against the
// LDS-DMA (counted vmcnt + barrier) and must issue `s_waitcnt lgkmcnt(N)` + __builtin_amdgcn_sched_barrier(0)
// before the first use of the result (the compiler does not know the value is still in flight).
__device__ __forceinline__ bf16x8 lds_read_tr_frag_async(uint32_t a0, uint32_t a1) {
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// async 16-byte global -> LDS copy; LDS destination = wave-uniform `lds_base` + lane*16
__device__ __forceinline__ void glds16(const void* gptr, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// XCD-aware, bijective block-id remap: consecutive logical ids land on the same XCD (8 XCDs,
// hardware round-robins blockIdx over XCDs).  cdna guide section 5.5 T1.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
