// L2 -> LDS copy rate of one workgroup per CU (buffer_load ... lds, 16 B per lane), all CUs at once: what can the operand DMA of a GEMM tile reach?
// Every workgroup (512 threads) re-reads a region of `rows` x 128 B (row stride `ld` bytes) `iters` times into a 64 KB LDS ring, `depth` requests per wave
// in flight (s_waitcnt vmcnt(depth - 1) before each new one).  mode 0: every workgroup has its own region (L2-resident after the first pass);
// mode 1: all workgroups read the SAME region (a weight tile shared by the CUs of an XCD).  Build: hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const char* base, int rows, int ld, long region_stride, int iters, unsigned long long* clk, int stream_rows = 0, const char* sbase = nullptr, long sstride = 0, int share = 1, int pf = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* reg = base + (long)blockIdx.x * region_stride;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)reg, 0, rows * ld, 0x00020000);
  __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(sbase + (long)(blockIdx.x / share) * sstride), 0, 0x7fffffff, 0x00020000);
  const char* sreg_ = sbase + (long)(blockIdx.x / share) * sstride; unsigned pfacc = 0;
  const int per_pass = rows / 64;                      // requests per wave per pass over the region (8 waves x 8 rows per request)
  const unsigned long long t0 = wall_clock64();
  int issued = 0;
  for (int it = 0; it < iters; ++it) {
    if (pf > 0 && tid >= 384 && tid - 384 < stream_rows && it + pf < iters && (blockIdx.x % share) == (it % share))
      pfacc += *reinterpret_cast<const volatile unsigned*>(sreg_ + (long)(tid - 384) * ld + (long)(it + pf) * 128);
    for (int q = 0; q < per_pass; ++q) {
      const int row = (q * 8 + wave) * 8 + (lane >> 3);
      const unsigned voff = (unsigned)row * ld + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
      if (row < stream_rows) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (__attribute__((address_space(3))) void*)(smem + ((issued & 7) * 8 + wave) * 1024), 16, voff, it * 128, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((issued & 7) * 8 + wave) * 1024), 16, voff, 0, 0, 0);
      ++issued;
      if (issued >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (pfacc == 0x12345u) clk[0] = 1;
  if (tid == 0) { clk[blockIdx.x * 2] = t0; clk[blockIdx.x * 2 + 1] = wall_clock64(); }
}
int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 416, ld = argc > 2 ? atoi(argv[2]) : 6144, iters = argc > 3 ? atoi(argv[3]) : 200;
  int cus = 256; hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); cus = pr.multiProcessorCount;
  char* buf; const long region = (long)rows * ld; hipMalloc(&buf, region * cus + 4096); hipMemset(buf, 1, region * cus + 4096);
  unsigned long long* clk; hipMalloc(&clk, cus * 16);
  std::vector<unsigned long long> h(cus * 2);
  for (int mode = 0; mode < 2; ++mode)
    for (int depth : {2, 4, 8, 16}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        const long stride = mode == 0 ? region : 0;
        if (depth == 2) hipLaunchKernelGGL(dma_kernel<2>, dim3(cus), dim3(512), 65536, 0, buf, rows, ld, stride, iters, clk);
        if (depth == 4) hipLaunchKernelGGL(dma_kernel<4>, dim3(cus), dim3(512), 65536, 0, buf, rows, ld, stride, iters, clk);
        if (depth == 8) hipLaunchKernelGGL(dma_kernel<8>, dim3(cus), dim3(512), 65536, 0, buf, rows, ld, stride, iters, clk);
        if (depth == 16) hipLaunchKernelGGL(dma_kernel<16>, dim3(cus), dim3(512), 65536, 0, buf, rows, ld, stride, iters, clk);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) {
          const double bytes = (double)(rows / 64) * 64 * 128 * iters;          // per workgroup
          printf("mode %d (%s) rows %d ld %d depth %2d: %.1f us, %.1f GB/s per CU, %.2f TB/s chip\n", mode, mode ? "shared region" : "own region", rows, ld, depth,
                 ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * cus / (ms * 1e-3) / 1e12);
        }
      }
    }
  // mode 2: the first 128 rows of every pass stream through a fresh [128 rows][iters x 128 B] panel of the workgroup (an A tile from HBM, ld = iters * 128), the rest is the shared region
  {
    const int srows = 128; const long sld = (long)iters * 128, sreg = srows * sld;
    char* sb; hipMalloc(&sb, sreg * cus + 4096); hipMemset(sb, 1, sreg * cus + 4096);
    for (int share : {1, 3}) for (int pf : {0, 4, 8}) for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(dma_kernel<8>, dim3(cus), dim3(512), 65536, 0, buf, rows, (int)sld, 0L, iters, clk, srows, sb, sreg, share, pf);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep == 1) { const double bytes = (double)(rows / 64) * 64 * 128 * iters;
        printf("mode 2 (128 of %d rows streamed from HBM, panel shared by %d workgroups, L2 prefetch %d passes ahead) depth 8: %.1f us, %.1f GB/s per CU, %.2f TB/s chip\n", rows, share, pf, ms * 1e3,
               bytes / (ms * 1e-3) / 1e9, bytes * cus / (ms * 1e-3) / 1e12); }
    }
  }
  return 0;
}
