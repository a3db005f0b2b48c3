// Multi-tensor AdamW + global gradient norm for the X^2-VLM step (gfx950): one launch over all ~570 parameter
// tensors instead of a Python loop of small ATen ops.
// Reference: optim.py:26-104 (param groups, transformers==4.12.5 AdamW(eps=1e-8, betas=(0.9,0.98), correct_bias=True)),
// accelerators/apex_ddp_accelerator.py:99-102 (clip_grad_norm_).  HF AdamW update, per element:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps) ; p -= lr*wd*p
// HBM-bound: 16 B/param read (p, g, m, v) + 12 B written.
#include "x2_common.h"

#define OPT_CHUNK 16384        // elements per workgroup
// stepscale = sqrt(1 - b2^t) / (1 - b1^t) with t = THIS tensor's step count (HF keeps state["step"] per parameter: a tensor
// without a gradient in some iterations, e.g. bbox_head on image-only steps, lags behind the others)
struct OptTensor { float* p; const float* g; float* m; float* v; long n; int group; int blk0; float stepscale; int pad; };

__device__ __forceinline__ int find_tensor(const OptTensor* tab, int nt, int blk) {
  int lo = 0, hi = nt - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].blk0 <= blk) lo = mid; else hi = mid - 1; }
  return lo;
}

// partial[blk] = sum of g^2 over the block's chunk
__global__ __launch_bounds__(256) void gradsq_kernel(const OptTensor* __restrict__ tab, int nt, float* partial) {
  __shared__ float sh[4];
  const int t = find_tensor(tab, nt, blockIdx.x);
  const OptTensor T = tab[t];
  const long off = (long)(blockIdx.x - T.blk0) * OPT_CHUNK;
  const long n = min((long)OPT_CHUNK, T.n - off);
  const float* g = T.g + off;
  float s = 0.f;
  if (T.g) {
    const long n4 = (((uintptr_t)g & 15) == 0) ? n / 4 : 0;
    for (long i = threadIdx.x; i < n4; i += 256) { const float4 v = *reinterpret_cast<const float4*>(g + i * 4); s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// out[0] = sqrt(sum partial), out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_)
__global__ __launch_bounds__(256) void gradnorm_final_kernel(const float* __restrict__ partial, int n, float max_norm, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nrm = sqrtf(sh[0] + sh[1] + sh[2] + sh[3]);
    out[0] = nrm;
    out[1] = max_norm > 0.f ? fminf(1.f, max_norm / (nrm + 1e-6f)) : 1.f;
  }
}
extern "C" int x2_grad_norm(const void* table, int ntensors, int nblocks, float max_norm, float* partial, float* out2, void* stream) {
  X2_REQUIRE(table && ntensors > 0 && nblocks > 0 && partial && out2, "x2_grad_norm: bad arguments");
  hipLaunchKernelGGL(gradsq_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)table, ntensors, partial);
  hipLaunchKernelGGL(gradnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblocks, max_norm, out2);
  return x2_check_launch("x2_grad_norm");
}

struct AdamHyper { float lr[16]; float wd[16]; float b1, b2, eps; };

__global__ __launch_bounds__(256) void adamw_kernel(const OptTensor* __restrict__ tab, int nt, AdamHyper h, const float* __restrict__ clip) {
  const int t = find_tensor(tab, nt, blockIdx.x);
  const OptTensor T = tab[t];
  if (!T.g) return;                                   // parameter without a gradient this step (bbox_head on image-only steps)
  const long off = (long)(blockIdx.x - T.blk0) * OPT_CHUNK;
  const long n = min((long)OPT_CHUNK, T.n - off);
  const float gs = clip ? clip[1] : 1.f;
  const float lr = h.lr[T.group], wd = h.wd[T.group];
  const float step = lr * T.stepscale;
  float* p = T.p + off; const float* g = T.g + off; float* m = T.m + off; float* v = T.v + off;
  const bool al = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const long n4 = al ? n / 4 : 0;
  for (long i = threadIdx.x; i < n4; i += 256) {
    float4 pp = *reinterpret_cast<float4*>(p + i * 4), mm = *reinterpret_cast<float4*>(m + i * 4), vv = *reinterpret_cast<float4*>(v + i * 4);
    const float4 gg = *reinterpret_cast<const float4*>(g + i * 4);
    float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; const float* G = &gg.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ge = G[e] * gs;
      M[e] = h.b1 * M[e] + (1.f - h.b1) * ge;
      V[e] = h.b2 * V[e] + (1.f - h.b2) * ge * ge;
      float x = P[e] - step * M[e] / (sqrtf(V[e]) + h.eps);
      P[e] = x - lr * wd * x;
    }
    *reinterpret_cast<float4*>(p + i * 4) = pp; *reinterpret_cast<float4*>(m + i * 4) = mm; *reinterpret_cast<float4*>(v + i * 4) = vv;
  }
  for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
    const float ge = g[i] * gs;
    const float me = h.b1 * m[i] + (1.f - h.b1) * ge, ve = h.b2 * v[i] + (1.f - h.b2) * ge * ge;
    m[i] = me; v[i] = ve;
    const float x = p[i] - step * me / (sqrtf(ve) + h.eps);
    p[i] = x - lr * wd * x;
  }
}
// table: ntensors x {p, g, m, v (pointers), n (long), group (int), blk0 (int), stepscale (float), pad} on the device;
// lr/wd: per group (<= 16)
extern "C" int x2_adamw_multi(const void* table, int ntensors, int nblocks, const float* lr, const float* wd, int ngroups, float b1,
                              float b2, float eps, const float* clip2, void* stream) {
  X2_REQUIRE(table && ntensors > 0 && nblocks > 0 && ngroups > 0 && ngroups <= 16, "x2_adamw_multi: bad arguments");
  AdamHyper h;
  for (int i = 0; i < 16; ++i) { h.lr[i] = i < ngroups ? lr[i] : 0.f; h.wd[i] = i < ngroups ? wd[i] : 0.f; }
  h.b1 = b1; h.b2 = b2; h.eps = eps;
  hipLaunchKernelGGL(adamw_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)table, ntensors, h, clip2);
  return x2_check_launch("x2_adamw_multi");
}
