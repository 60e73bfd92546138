// CTC greedy decode and generic row argmax (used by the
// Transformer decoder's LM head).
//
//   row_argmax : one wave64 per (sample, time) row of logits [R, V]; online
//                max / argmax / sum-exp with 16-byte vector loads, so the max
//                softmax probability comes out of the same pass.
//   ctc_collapse: one block per sample; drops blanks and repeats with a block
//                prefix sum, writes compacted ids, length and mean confidence.
#include "common.h"

template <typename T>
__device__ __forceinline__ float ld_logit(const T* p, long long i);
template <>
__device__ __forceinline__ float ld_logit<bf16_t>(const bf16_t* p, long long i) { return (float)p[i]; }
template <>
__device__ __forceinline__ float ld_logit<float>(const float* p, long long i) { return p[i]; }

template <typename T>
__global__ __launch_bounds__(256) void row_argmax_kernel(const T* __restrict__ logits, int* __restrict__ idx,
                                                         float* __restrict__ prob, long long R, int V,
                                                         long long row_stride) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const T* p = logits + row * row_stride;
  // finite sentinel: the library is built with -ffast-math (no infinities); idle lanes carry (m = -1e30, s = 0)
  float m = -1e30f, s = 0.f;
  int am = 0;
  if (sizeof(T) == 2 && (V % 8 == 0) && (row_stride % 8 == 0)) {
    const u32x4* pv = reinterpret_cast<const u32x4*>(p);
    for (int c8 = lane; c8 < V / 8; c8 += 64) {
      u32x4 raw = pv[c8];
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[2 * q] = bf16_bits_to_f32(raw[q] & 0xffffu);
        v[2 * q + 1] = bf16_bits_to_f32(raw[q] >> 16);
      }
      float lm = v[0]; int li = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) if (v[q] > lm) { lm = v[q]; li = q; }
      if (lm > m) { s *= __expf(m - lm); m = lm; am = c8 * 8 + li; }
#pragma unroll
      for (int q = 0; q < 8; ++q) s += __expf(v[q] - m);
    }
  } else {
    for (int c = lane; c < V; c += 64) {
      float v = ld_logit<T>(p, c);
      if (v > m) { s *= __expf(m - v); m = v; am = c; }
      s += __expf(v - m);
    }
  }
  // wave reduction of (m, am, s); ties -> smallest index (matches torch.argmax on first occurrence)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float om = __shfl_xor(m, o, 64);
    int oa = __shfl_xor(am, o, 64);
    float os = __shfl_xor(s, o, 64);
    float nm = fmaxf(m, om);
    float ns = s * __expf(m - nm) + os * __expf(om - nm);
    if (om > m || (om == m && oa < am)) am = oa;
    m = nm; s = ns;
  }
  if (lane == 0) {
    idx[row] = am;
    prob[row] = 1.f / s;   // exp(max - max) / sum exp(x - max)
  }
}

// idx/prob: [N, T]; lens: [N] valid time steps (or nullptr -> T); out_ids: [N, T] (padded with -1)
__global__ __launch_bounds__(256) void ctc_collapse_kernel(const int* __restrict__ idx, const float* __restrict__ prob,
                                                           const int* __restrict__ lens,
                                                           const long long* __restrict__ row_off,
                                                           int* __restrict__ out_ids,
                                                           int* __restrict__ out_len, float* __restrict__ out_conf,
                                                           int T, int blank) {
  __shared__ int s_cnt[256];
  __shared__ float s_p[256];
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int len = lens ? min(lens[n], T) : T;
  const int per = (T + 255) / 256;
  const int t0 = tid * per, t1 = min(t0 + per, len);
  // ragged input: sequence n starts at row_off[n] of the packed idx / prob vectors (output stays [N, T])
  const long long in0 = row_off ? row_off[n] : (long long)n * T;
  const int* row = idx + in0;
  const float* prow = prob + in0;
  int cnt = 0;
  float ps = 0.f;
  for (int t = t0; t < t1; ++t) {
    int v = row[t];
    bool keep = v != blank && (t == 0 || v != row[t - 1]);
    if (keep) { ++cnt; ps += prow[t]; }
  }
  s_cnt[tid] = cnt;
  s_p[tid] = ps;
  __syncthreads();
  // inclusive scan (Hillis-Steele) over 256 counts
  for (int o = 1; o < 256; o <<= 1) {
    int v = tid >= o ? s_cnt[tid - o] : 0;
    float pv = tid >= o ? s_p[tid - o] : 0.f;
    __syncthreads();
    s_cnt[tid] += v;
    s_p[tid] += pv;
    __syncthreads();
  }
  int pos = s_cnt[tid] - cnt;
  int* orow = out_ids + (long long)n * T;
  for (int t = t0; t < t1; ++t) {
    int v = row[t];
    bool keep = v != blank && (t == 0 || v != row[t - 1]);
    if (keep) orow[pos++] = v;
  }
  const int total = s_cnt[255];
  for (int t = total + tid; t < T; t += 256) orow[t] = -1;
  if (tid == 0) {
    out_len[n] = total;
    out_conf[n] = total > 0 ? s_p[255] / (float)total : 0.f;
  }
}

extern "C" int dlamd_row_argmax(const void* logits, int is_f32, void* idx, void* prob, long long R, int V,
                                long long row_stride, hipStream_t stream) {
  if (R <= 0) return 0;
  long long nb = (R + 3) / 4;
  if (nb > 0x7fffffffLL || V <= 0) return -1;
  if (is_f32)
    hipLaunchKernelGGL(row_argmax_kernel<float>, dim3((unsigned)nb), dim3(256), 0, stream, (const float*)logits,
                       (int*)idx, (float*)prob, R, V, row_stride);
  else
    hipLaunchKernelGGL(row_argmax_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, stream, (const bf16_t*)logits,
                       (int*)idx, (float*)prob, R, V, row_stride);
  return DLAMD_CHECK_LAUNCH();
}

extern "C" int dlamd_ctc_collapse(const void* idx, const void* prob, const void* lens, const void* row_off,
                                  void* out_ids, void* out_len, void* out_conf, int N, int T, int blank,
                                  hipStream_t stream) {
  if (N <= 0) return 0;
  if (T <= 0) return -1;
  if (row_off && !lens) return -2;
  hipLaunchKernelGGL(ctc_collapse_kernel, dim3(N), dim3(256), 0, stream, (const int*)idx, (const float*)prob,
                     (const int*)lens, (const long long*)row_off, (int*)out_ids, (int*)out_len, (float*)out_conf, T, blank);
  return DLAMD_CHECK_LAUNCH();
}
