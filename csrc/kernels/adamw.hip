// Fused multi-tensor AdamW over a flat parameter bucket.
//
// Parameters, gradients and both moments of a whole DP bucket are contiguous
// (parallel/ddp.py FlatBuckets), so one launch updates tens of MB with 16-byte
// vector accesses: 4 streams read (p, g, m, v), 3 written (p, m, v) [+ bf16 shadow].
// Pure HBM streaming: grid = enough 256-thread blocks to cover the chip several times.
#include "common.h"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ shadow, long long n4, long long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    const float* __restrict__ gscale) {
  const float gs = gscale ? *gscale : 1.f;
  const float step = lr / bc1;
  const float inv_bc2_sqrt = rsqrtf(bc2);
  const float decay = 1.f - lr * wd;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = gg[k] * gs;
      mm[k] = b1 * mm[k] + (1.f - b1) * gk;
      vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
      float denom = sqrtf(vv[k]) * inv_bc2_sqrt + eps;
      pp[k] = pp[k] * decay - step * (mm[k] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (shadow) {
      bf16x4 s = {(bf16_t)pp[0], (bf16_t)pp[1], (bf16_t)pp[2], (bf16_t)pp[3]};
      reinterpret_cast<bf16x4*>(shadow)[i] = s;
    }
  }
  // scalar tail (n not a multiple of 4)
  if (blockIdx.x == 0) {
    for (long long i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      float gk = g[i] * gs;
      float mk = b1 * m[i] + (1.f - b1) * gk;
      float vk = b2 * v[i] + (1.f - b2) * gk * gk;
      float denom = sqrtf(vk) * inv_bc2_sqrt + eps;
      float pk = p[i] * decay - step * (mk / denom);
      p[i] = pk; m[i] = mk; v[i] = vk;
      if (shadow) shadow[i] = (bf16_t)pk;
    }
  }
}

// p, g, m, v: fp32 [n] (16-byte aligned); shadow: optional bf16 [n] copy of the updated parameters;
// bc1 = 1 - beta1^t, bc2 = 1 - beta2^t; gscale: optional device scalar multiplied into every gradient
// (gradient clipping / loss-scale) so no separate pass over the gradients is needed.
extern "C" int dlamd_adamw(void* p, const void* g, void* m, void* v, void* shadow, long long n, float lr, float b1,
                           float b2, float eps, float wd, float bc1, float bc2, const void* gscale,
                           hipStream_t stream) {
  if (n <= 0) return 0;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return -1;
  if (shadow && ((uintptr_t)shadow & 7)) return -2;
  long long n4 = n / 4;
  long long blocks = (n4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (float*)p, (const float*)g, (float*)m,
                     (float*)v, (bf16_t*)shadow, n4, n, lr, b1, b2, eps, wd, bc1, bc2, (const float*)gscale);
  return DLAMD_CHECK_LAUNCH();
}

// sum of squares of a flat fp32 buffer, accumulated into *out (caller zeroes it): global grad-norm for clipping.
// Deterministic two-stage sum of squares: per-block partials, then ONE block folds them in a fixed order and adds the
// total to *out with a plain read-modify-write (launches on a stream are ordered).  A float atomicAdd per block would
// make the gradient norm - and through the clipping scale every parameter - depend on the order the blocks retire,
// which lets data-parallel ranks that hold bitwise-identical gradients drift apart.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float x = g[i];
    acc += x * x;
  }
  acc = wave_reduce_sum(acc);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partial[i];
  acc = wave_reduce_sum(acc);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (s[0] + s[1]) + (s[2] + s[3]);
}

static float* sumsq_scratch() {
  static float* buf[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!buf[dev] && hipMalloc((void**)&buf[dev], 1024 * sizeof(float)) != hipSuccess) buf[dev] = nullptr;
  return buf[dev];
}

extern "C" int dlamd_sumsq(const void* g, long long n, void* out, hipStream_t stream) {
  if (n <= 0) return 0;
  long long blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > 1024) blocks = 1024;
  float* partial = sumsq_scratch();     // one optimizer stream per process: the scratch is reused call after call
  if (!partial) return -2;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)g, n, partial);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, (int)blocks, (float*)out);
  return DLAMD_CHECK_LAUNCH();
}
