// Hardware probe: pins down gfx950 instruction semantics the kernels rely on.
// Build: hipcc --offload-arch=gfx950 -O2 probes/probe_isa.hip -o probes/probe_isa
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__global__ void k_trread(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v;
  uint32_t addr = (uint32_t)(uintptr_t)(&lds[0]) + l * 8;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
// second form: per-lane addresses with a row stride of 64 elements (128 B): lane (r*4+c4) of each
// 16-lane group g points at row g*4+r ... to test "row stride is free"
__global__ void k_trread2(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int g = l >> 4, i = l & 15, r = i >> 2, c4 = i & 3;
  // rows k = g*8 + r (stride 64 elements), cols 4*c4..4*c4+3
  uint32_t addr = (uint32_t)(uintptr_t)(&lds[0]) + ((g * 8 + r) * 64 + c4 * 4) * 2;
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
__device__ inline short f2bf(float f) { uint32_t u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return (short)(u >> 16); }
// C = A(16x32) * B(32x16); A[i][k] = i*100+k (small ints exact in bf16? use i + k/64.), test with A=I-ish
__global__ void k_mfma(float* out, const float* A, const float* B) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = f2bf(A[(l & 15) * 32 + (l >> 4) * 8 + j]);        // A[row=l&15][k=(l>>4)*8+j]
    b[j] = f2bf(B[((l >> 4) * 8 + j) * 16 + (l & 15)]);      // B[k][n=l&15]
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];   // row=(l>>4)*4+r, col=l&15
}
__global__ void k_glds(uint32_t* out, const uint32_t* src) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2048];
  int l = threadIdx.x;
  for (int i = l; i < 2048; i += 64) lds[i] = 0xdeadbeef;
  __syncthreads();
  // each lane sources 16 B from src + (63-l)*4 dwords (reversed) ; LDS base uniform = &lds[256]
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (63 - l) * 4),
                                   (__attribute__((address_space(3))) void*)(&lds[256]), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 2048; i += 64) out[i] = lds[i];
}
__global__ void k_cvtpk(uint32_t* out, const float* in) {
  int l = threadIdx.x; uint32_t r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(in[2 * l]), "v"(in[2 * l + 1]));
  out[l] = r;
}
__global__ void k_copy(float4* __restrict__ d, const float4* __restrict__ s, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) d[i] = s[i];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s arch %s CUs %d clock %d kHz mem %.1f GB L2 %d\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1e9, p.l2CacheSize);
  uint16_t* d16; CK(hipMalloc(&d16, 4096)); std::vector<uint16_t> h16(256);
  k_trread<<<1, 64>>>(d16); CK(hipMemcpy(h16.data(), d16, 512, hipMemcpyDeviceToHost));
  printf("TRREAD lane-linear (addr=lane*8B), lds[i]=i: per lane 4 values\n");
  for (int l = 0; l < 64; ++l) { printf("L%02d:%4d %4d %4d %4d |", l, h16[l*4], h16[l*4+1], h16[l*4+2], h16[l*4+3]); if (l % 4 == 3) printf("\n"); }
  k_trread2<<<1, 64>>>(d16); CK(hipMemcpy(h16.data(), d16, 512, hipMemcpyDeviceToHost));
  printf("TRREAD2 rows stride 64: expect lane(g,i) -> lds[(g*8+j)*64 + i], j=0..3\n");
  int bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int e = ((l >> 4) * 8 + j) * 64 + (l & 15); if (h16[l*4+j] != e) ++bad; }
  printf("TRREAD2 mismatches %d\n", bad);
  if (bad) for (int l = 0; l < 64; ++l) { printf("L%02d:%4d %4d %4d %4d |", l, h16[l*4], h16[l*4+1], h16[l*4+2], h16[l*4+3]); if (l % 4 == 3) printf("\n"); }
  // MFMA layout
  std::vector<float> A(16 * 32), B(32 * 16), C(256), R(256);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) B[k * 16 + n] = (float)((k * 5 + n * 13) % 7 - 3);
  for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i*32+k] * B[k*16+n]; R[i*16+n] = s; }
  float *dA, *dB, *dC; CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dC, 1024));
  CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
  k_mfma<<<1, 64>>>(dC, dA, dB); CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
  bad = 0; for (int i = 0; i < 256; ++i) if (C[i] != R[i]) ++bad; printf("MFMA16x16x32 layout mismatches %d\n", bad);
  // glds
  uint32_t *ds, *dd; CK(hipMalloc(&ds, 1024)); CK(hipMalloc(&dd, 8192)); std::vector<uint32_t> hs(256), hd(2048);
  for (int i = 0; i < 256; ++i) hs[i] = i; CK(hipMemcpy(ds, hs.data(), 1024, hipMemcpyHostToDevice));
  k_glds<<<1, 64>>>(dd, ds); CK(hipMemcpy(hd.data(), dd, 8192, hipMemcpyDeviceToHost));
  bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) if (hd[256 + l * 4 + j] != (uint32_t)((63 - l) * 4 + j)) ++bad;
  int touched = 0; for (int i = 0; i < 2048; ++i) if (hd[i] != 0xdeadbeef) ++touched;
  printf("GLDS16 lane-linear dest mismatches %d, touched dwords %d (expect 256)\n", bad, touched);
  float* df; uint32_t* du; CK(hipMalloc(&df, 512)); CK(hipMalloc(&du, 256)); std::vector<float> hf(128); for (int i = 0; i < 128; ++i) hf[i] = 1.0f + i * 0.00390625f;
  CK(hipMemcpy(df, hf.data(), 512, hipMemcpyHostToDevice)); k_cvtpk<<<1, 64>>>(du, df); std::vector<uint32_t> hu(64); CK(hipMemcpy(hu.data(), du, 256, hipMemcpyDeviceToHost));
  printf("CVTPK lane1: %08x (lo should be bf16(%f) hi bf16(%f))\n", hu[1], hf[2], hf[3]);
  // copy bandwidth
  size_t n = (size_t)1 << 30; float4 *s4, *d4; CK(hipMalloc(&s4, n)); CK(hipMalloc(&d4, n)); CK(hipMemset(s4, 1, n));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) k_copy<<<2048, 256>>>(d4, s4, n / 16);
  hipEventRecord(e0); for (int it = 0; it < 10; ++it) k_copy<<<2048, 256>>>(d4, s4, n / 16); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); printf("COPY 1GiB x10: %.3f ms each -> %.2f TB/s (r+w)\n", ms / 10, 2.0 * n / (ms / 10 * 1e-3) / 1e12);
  return 0;
}
