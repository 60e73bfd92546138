// Batch normalisation for TRAINING on NHWC bf16 activations, fused with the activation that follows it.
//
//   forward : y = act(gamma * (z - mean) * invstd + beta), batch statistics per channel over all N*H*W pixels
//   backward: dz = gamma * invstd * (dyh - sum(dyh) / P - xhat * sum(dyh * xhat) / P),  dyh = dy * act'(.)
//
// Both directions are two memory-bound passes: a per-channel reduction (every workgroup reduces a slab of pixels for
// all channels in registers, combines its row groups through LDS and writes one row of partial sums; a tiny second
// kernel adds the rows) and an
// elementwise pass with 16-byte accesses.  The activation's mask is recomputed from z in the backward pass, so only
// z (which the convolution's weight gradient keeps anyway), mean and invstd are saved - the separate ReLU forward /
// backward kernels and their tensors disappear.  Channels are a multiple of 8 and at most 512; act: none or ReLU.
#include "common.h"

static constexpr int BN_ROWS_PER_WG = 512;   // pixels per workgroup of the reduction passes

__device__ __forceinline__ void bn_unpack8(const u32x4 v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf16_bits_to_f32(v[i] & 0xffffu);
    f[2 * i + 1] = bf16_bits_to_f32(v[i] >> 16);
  }
}

__device__ __forceinline__ u32x4 bn_pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2 p = {(bf16_t)f[2 * i], (bf16_t)f[2 * i + 1]};
    v[i] = __builtin_bit_cast(uint32_t, p);
  }
  return v;
}

// Combine the 8-channel partial sums of the row groups of a workgroup and write them to this workgroup's partial row.
// part: this thread's sums for its 8 channels (two quantities a, b); cc = chunks per row (C / 8); group = tid / cc.
__device__ __forceinline__ void bn_block_reduce(const float (&a)[8], const float (&b)[8], float* lds, int cc, int chunk,
                                                int group, int groups, float* ga, float* gb) {
  // lds: [groups][cc][16]
  float* slot = lds + ((long long)group * cc + chunk) * 16;
#pragma unroll
  for (int j = 0; j < 8; ++j) { slot[j] = a[j]; slot[8 + j] = b[j]; }
  __syncthreads();
  if (group == 0) {
    float sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sa[j] = 0.f; sb[j] = 0.f; }
    for (int g = 0; g < groups; ++g) {
      const float* s = lds + ((long long)g * cc + chunk) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j) { sa[j] += s[j]; sb[j] += s[8 + j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ga[chunk * 8 + j] = sa[j];
      gb[chunk * 8 + j] = sb[j];
    }
  }
}

// sums[i] = sum over workgroups of partial[g][i], i < 2 C (atomics on 2 C addresses from thousands of workgroups
// serialise in one cache line; a second pass over a few hundred partial rows costs microseconds)
__global__ __launch_bounds__(1024) void bn_finish_kernel(const float* __restrict__ partial, float* __restrict__ sums, int G, int n) {
  // 16 output sums per workgroup, 64 threads per sum: each adds every 64th partial row, then a wave-wide tree
  __shared__ float lds[1024];
  const int il = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + il;
  float s = 0.f;
  if (i < n)
    for (int g = part; g < G; g += 64) s += partial[(long long)g * n + i];
  lds[threadIdx.x] = s;
  __syncthreads();
  for (int h = 32; h > 0; h >>= 1) {
    if (part < h) lds[threadIdx.x] += lds[threadIdx.x + h * 16];
    __syncthreads();
  }
  if (part == 0 && i < n) sums[i] = lds[il];
}

// partial[g][0][C] = sum z, partial[g][1][C] = sum z^2 over the slabs of workgroup g
__global__ __launch_bounds__(256) void bn_stats_kernel(const bf16_t* __restrict__ z, float* __restrict__ partial, long long P, int C) {
  __shared__ float lds[256 * 16];
  const int cc = C >> 3, tid = threadIdx.x;
  const int groups = 256 / cc;
  const int chunk = tid % cc, group = tid / cc;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = 0.f; b[j] = 0.f; }
  for (long long r0 = (long long)blockIdx.x * BN_ROWS_PER_WG; r0 < P; r0 += (long long)gridDim.x * BN_ROWS_PER_WG) {
    const long long r1 = min(r0 + BN_ROWS_PER_WG, P);
    for (long long r = r0 + group; r < r1; r += groups) {
      float f[8];
      bn_unpack8(*reinterpret_cast<const u32x4*>(z + r * C + chunk * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { a[j] += f[j]; b[j] += f[j] * f[j]; }
    }
  }
  float* row = partial + (long long)blockIdx.x * 2 * C;
  bn_block_reduce(a, b, lds, cc, chunk, group, groups, row, row + C);
}

// One launch of C / 8 threads ... folded into the apply kernels: every thread derives scale / shift of its 8 channels.
__device__ __forceinline__ void bn_channel_params(const float* __restrict__ sums, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, long long P, int C, float eps, int c0,
                                                  float (&mean)[8], float (&invstd)[8], float (&scale)[8], float (&shift)[8]) {
  const float inv_p = 1.f / (float)P;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float m = sums[c0 + j] * inv_p;
    const float var = fmaxf(sums[C + c0 + j] * inv_p - m * m, 0.f);
    const float is = rsqrtf(var + eps);
    mean[j] = m;
    invstd[j] = is;
    scale[j] = (gamma ? gamma[c0 + j] : 1.f) * is;
    shift[j] = (beta ? beta[c0 + j] : 0.f) - m * scale[j];
  }
}

// y = act(z * scale + shift); block 0 also writes mean / invstd (saved for backward) and updates the running statistics.
// A thread keeps one 8-channel chunk (its scale / shift are computed once) and walks down the rows of its slabs.
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_t* __restrict__ z, const float* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16_t* __restrict__ y, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var, long long P, int C, float eps,
                                                       float momentum, int act) {
  const int cc = C >> 3, tid = threadIdx.x;
  const int groups = 256 / cc;
  const int chunk = tid % cc, group = tid / cc;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      const float m = sums[c] / (float)P;
      const float var = fmaxf(sums[C + c] / (float)P - m * m, 0.f);
      save_mean[c] = m;
      save_invstd[c] = rsqrtf(var + eps);
      if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
      if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * ((float)P / fmaxf((float)P - 1.f, 1.f));
    }
  }
  float mean[8], invstd[8], scale[8], shift[8];
  bn_channel_params(sums, gamma, beta, P, C, eps, chunk * 8, mean, invstd, scale, shift);
  for (long long r0 = (long long)blockIdx.x * BN_ROWS_PER_WG; r0 < P; r0 += (long long)gridDim.x * BN_ROWS_PER_WG) {
    const long long r1 = min(r0 + BN_ROWS_PER_WG, P);
    for (long long r = r0 + group; r < r1; r += groups) {
      float f[8];
      bn_unpack8(*reinterpret_cast<const u32x4*>(z + r * C + chunk * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = f[j] * scale[j] + shift[j];
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        f[j] = v;
      }
      *reinterpret_cast<u32x4*>(y + r * C + chunk * 8) = bn_pack8(f);
    }
  }
}

// partial[g][0][C] = sum dyh, partial[g][1][C] = sum dyh * xhat   (dyh = dy where the activation passed, else 0)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ partial, long long P, int C, int act) {
  __shared__ float lds[256 * 16];
  const int cc = C >> 3, tid = threadIdx.x;
  const int groups = 256 / cc;
  const int chunk = tid % cc, group = tid / cc;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = 0.f; b[j] = 0.f; }
  {
    float m[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = chunk * 8 + j;
      m[j] = mean[c];
      is[j] = invstd[c];
      sc[j] = (gamma ? gamma[c] : 1.f) * is[j];
      sh[j] = (beta ? beta[c] : 0.f) - m[j] * sc[j];
    }
    for (long long r0 = (long long)blockIdx.x * BN_ROWS_PER_WG; r0 < P; r0 += (long long)gridDim.x * BN_ROWS_PER_WG) {
    const long long r1 = min(r0 + BN_ROWS_PER_WG, P);
    for (long long r = r0 + group; r < r1; r += groups) {
      float fz[8], fd[8];
      bn_unpack8(*reinterpret_cast<const u32x4*>(z + r * C + chunk * 8), fz);
      bn_unpack8(*reinterpret_cast<const u32x4*>(dy + r * C + chunk * 8), fd);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = (act == ACT_RELU && fz[j] * sc[j] + sh[j] <= 0.f) ? 0.f : fd[j];
        a[j] += g;
        b[j] += g * (fz[j] - m[j]) * is[j];
      }
    }
    }
  }
  float* row = partial + (long long)blockIdx.x * 2 * C;
  bn_block_reduce(a, b, lds, cc, chunk, group, groups, row, row + C);
}

// dz = gamma * invstd * (dyh - sum_dyh / P - xhat * sum_dyh_xhat / P)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ sums, bf16_t* __restrict__ dz,
                                                           long long P, int C, int act) {
  const int cc = C >> 3, tid = threadIdx.x;
  const int groups = 256 / cc;
  const int chunk = tid % cc, group = tid / cc;
  const float inv_p = 1.f / (float)P;
  float m[8], is[8], sc[8], sh[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = chunk * 8 + j;
    m[j] = mean[c];
    is[j] = invstd[c];
    sc[j] = (gamma ? gamma[c] : 1.f) * is[j];
    sh[j] = (beta ? beta[c] : 0.f) - m[j] * sc[j];
    k1[j] = sums[c] * inv_p;
    k2[j] = sums[C + c] * inv_p;
  }
  for (long long r0 = (long long)blockIdx.x * BN_ROWS_PER_WG; r0 < P; r0 += (long long)gridDim.x * BN_ROWS_PER_WG) {
    const long long r1 = min(r0 + BN_ROWS_PER_WG, P);
    for (long long r = r0 + group; r < r1; r += groups) {
      float fz[8], fd[8];
      bn_unpack8(*reinterpret_cast<const u32x4*>(z + r * C + chunk * 8), fz);
      bn_unpack8(*reinterpret_cast<const u32x4*>(dy + r * C + chunk * 8), fd);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = (act == ACT_RELU && fz[j] * sc[j] + sh[j] <= 0.f) ? 0.f : fd[j];
        const float xh = (fz[j] - m[j]) * is[j];
        fd[j] = sc[j] * (g - k1[j] - xh * k2[j]);
      }
      *reinterpret_cast<u32x4*>(dz + r * C + chunk * 8) = bn_pack8(fd);
    }
  }
}

static bool bn_shape_ok(long long P, int C) { return P > 0 && C >= 8 && C <= 512 && C % 8 == 0 && 256 % (C / 8) == 0; }

static constexpr int BN_MAX_WGS = 512;     // workgroups of a reduction pass = rows of the partial-sum scratch

// z, y: [P, C] bf16; sums: [2, C] fp32 out; partial: scratch of BN_MAX_WGS * 2 * C floats; save_mean / save_invstd: [C]
// fp32 out; run_mean / run_var: [C] fp32 updated in place (or nullptr).
extern "C" int dlamd_bn_train_fwd(const void* z, const float* gamma, const float* beta, void* y, float* sums, float* partial,
                                  float* save_mean,
                                  float* save_invstd, float* run_mean, float* run_var, long long P, int C, float eps,
                                  float momentum, int act, hipStream_t stream) {
  if (!z || !y || !sums || !partial || !save_mean || !save_invstd || !bn_shape_ok(P, C)) return -1;
  if (act != ACT_NONE && act != ACT_RELU) return -2;
  long long wgs = (P + BN_ROWS_PER_WG - 1) / BN_ROWS_PER_WG;
  if (wgs > BN_MAX_WGS) wgs = BN_MAX_WGS;
  const long long chunks = P * (C / 8);
  if ((chunks + 255) / 256 > 0x7fffffffLL) return -3;
  hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, (const bf16_t*)z, partial, P, C);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((2 * C + 15) / 16), dim3(1024), 0, stream, (const float*)partial, sums, (int)wgs, 2 * C);
  long long awgs = (P + BN_ROWS_PER_WG - 1) / BN_ROWS_PER_WG;
  if (awgs > 4096) awgs = 4096;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)awgs), dim3(256), 0, stream, (const bf16_t*)z, sums, gamma,
                     beta, (bf16_t*)y, save_mean, save_invstd, run_mean, run_var, P, C, eps, momentum, act);
  return DLAMD_CHECK_LAUNCH();
}

// sums: [2, C] fp32 out: sums[0] = d beta, sums[1] = d gamma; partial: scratch of BN_MAX_WGS * 2 * C floats.
extern "C" int dlamd_bn_train_bwd(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, float* sums, float* partial, void* dz, long long P, int C, int act,
                                  hipStream_t stream) {
  if (!dy || !z || !mean || !invstd || !sums || !partial || !dz || !bn_shape_ok(P, C)) return -1;
  if (act != ACT_NONE && act != ACT_RELU) return -2;
  long long wgs = (P + BN_ROWS_PER_WG - 1) / BN_ROWS_PER_WG;
  if (wgs > BN_MAX_WGS) wgs = BN_MAX_WGS;
  const long long chunks = P * (C / 8);
  if ((chunks + 255) / 256 > 0x7fffffffLL) return -3;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)z, mean,
                     invstd, gamma, beta, partial, P, C, act);
  hipLaunchKernelGGL(bn_finish_kernel, dim3((2 * C + 15) / 16), dim3(1024), 0, stream, (const float*)partial, sums, (int)wgs, 2 * C);
  long long awgs = (P + BN_ROWS_PER_WG - 1) / BN_ROWS_PER_WG;
  if (awgs > 4096) awgs = 4096;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)awgs), dim3(256), 0, stream, (const bf16_t*)dy,
                     (const bf16_t*)z, mean, invstd, gamma, beta, sums, (bf16_t*)dz, P, C, act);
  return DLAMD_CHECK_LAUNCH();
}
