// CTC loss forward + gradient w.r.t. the logits.
//
//   1. ctc_lse_kernel     log-sum-exp of every (sample, time) row of the logits:
//                         one wave64 per row, fully parallel over N*T rows.
//   2. ctc_alpha_beta_kernel   one workgroup per sample; the S = 2L+1 extended-label
//                         states are spread over the lanes, time is the sequential
//                         loop.  alpha is kept in global scratch [T, S] (L2-resident),
//                         beta lives in LDS (double buffered).  The gradient
//                             dL/dz[t,v] = softmax[t,v] - sum_{s: l'_s = v} exp(alpha_t(s) + beta_t(s) + nll - logp[t,v])
//                         is produced in the same backward sweep: per time step the
//                         occupancy of every class is accumulated with LDS float
//                         atomics in the linear domain (shifted by the step's max),
//                         so repeated labels and the L+1 blanks need no sorting.
// Log-domain maths in fp32; logits may be bf16 or fp32; the gradient is fp32.
#include "common.h"

#define CTC_NEG_INF (-1e30f)

__device__ __forceinline__ float log_add(float a, float b) {
  float mx = fmaxf(a, b), mn = fminf(a, b);
  if (mx <= CTC_NEG_INF) return CTC_NEG_INF;
  return mx + log1pf(__expf(mn - mx));
}

template <typename LT>
__global__ __launch_bounds__(256) void ctc_lse_kernel(const LT* __restrict__ logits, float* __restrict__ lse,
                                                      long long rows, int V) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const LT* r = logits + row * V;
  float mx = CTC_NEG_INF;
  for (int v = lane; v < V; v += 64) mx = fmaxf(mx, (float)r[v]);
  mx = wave_reduce_max(mx);
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += __expf((float)r[v] - mx);
  s = wave_reduce_sum(s);
  if (lane == 0) lse[row] = mx + __logf(s);
}

template <typename LT>
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(
    const LT* __restrict__ logits, const float* __restrict__ lse, const int* __restrict__ targets,
    const int* __restrict__ tgt_off, const int* __restrict__ tgt_len, const int* __restrict__ in_len,
    float* __restrict__ alpha_ws, float* __restrict__ loss, float* __restrict__ grad, int T, int V, int Smax,
    int blank, int zero_infinity) {
  extern __shared__ float smem[];
  // layout: lab[Smax] (int) | a0[Smax] | a1[Smax] | occ[V] | red[8]
  int* lab = reinterpret_cast<int*>(smem);
  float* a0 = smem + Smax;
  float* a1 = a0 + Smax;
  float* occ = a1 + Smax;
  float* red = occ + V;
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int L = tgt_len[n];
  const int S = 2 * L + 1;
  const int Tn = min(in_len[n], T);
  const LT* lg = logits + (long long)n * T * V;
  const float* ls = lse + (long long)n * T;
  float* aw = alpha_ws + (long long)n * T * Smax;
  float* gr = grad + (long long)n * T * V;
  for (int s = tid; s < S; s += 256) lab[s] = (s & 1) ? targets[tgt_off[n] + (s >> 1)] : blank;
  __syncthreads();

  auto logp = [&](int t, int c) { return (float)lg[(long long)t * V + c] - ls[t]; };

  // ---------------- alpha ----------------
  bool feasible = Tn > 0 && S <= Smax;
  if (feasible) {
    for (int s = tid; s < S; s += 256) {
      float a = CTC_NEG_INF;
      if (s == 0) a = logp(0, blank);
      else if (s == 1) a = logp(0, lab[1]);
      a0[s] = a;
      aw[s] = a;
    }
    __syncthreads();
    float* cur = a0;
    float* nxt = a1;
    for (int t = 1; t < Tn; ++t) {
      for (int s = tid; s < S; s += 256) {
        float a = cur[s];
        if (s >= 1) a = log_add(a, cur[s - 1]);
        if (s >= 2 && (s & 1) && lab[s] != lab[s - 2]) a = log_add(a, cur[s - 2]);
        a = (a <= CTC_NEG_INF) ? CTC_NEG_INF : a + logp(t, lab[s]);
        nxt[s] = a;
        aw[(long long)t * Smax + s] = a;
      }
      __syncthreads();
      float* tmp = cur; cur = nxt; nxt = tmp;
    }
    float ll = cur[S - 1];
    if (S > 1) ll = log_add(ll, cur[S - 2]);
    if (tid == 0) red[0] = ll;
    __syncthreads();
  }
  const float ll = feasible ? red[0] : CTC_NEG_INF;
  const bool inf_loss = !(ll > CTC_NEG_INF);
  const float nll = -ll;
  if (tid == 0) loss[n] = inf_loss ? (zero_infinity ? 0.f : -CTC_NEG_INF) : nll;  // 1e30 stands for +inf (fast-math build)
  __syncthreads();

  // ---------------- beta + gradient ----------------
  // padded time steps and infeasible samples get a zero gradient
  for (long long i = (long long)Tn * V + tid; i < (long long)T * V; i += 256) gr[i] = 0.f;
  if (inf_loss) {
    for (long long i = tid; i < (long long)Tn * V; i += 256) gr[i] = 0.f;
    return;
  }
  float* cur = a0;
  float* nxt = a1;
  for (int t = Tn - 1; t >= 0; --t) {
    // beta_t(s), including the emission at t
    for (int s = tid; s < S; s += 256) {
      float b;
      if (t == Tn - 1) {
        b = (s == S - 1 || s == S - 2) ? 0.f : CTC_NEG_INF;
      } else {
        b = cur[s];
        if (s + 1 < S) b = log_add(b, cur[s + 1]);
        if (s + 2 < S && (s & 1) && lab[s] != lab[s + 2]) b = log_add(b, cur[s + 2]);
      }
      b = (b <= CTC_NEG_INF) ? CTC_NEG_INF : b + logp(t, lab[s]);
      nxt[s] = b;
    }
    for (int v = tid; v < V; v += 256) occ[v] = 0.f;
    __syncthreads();
    // max over s of alpha + beta (shift for the linear-domain accumulation)
    float mx = CTC_NEG_INF;
    for (int s = tid; s < S; s += 256) mx = fmaxf(mx, aw[(long long)t * Smax + s] + nxt[s]);
    mx = wave_reduce_max(mx);
    if ((tid & 63) == 0) red[1 + (tid >> 6)] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[1], red[2]), fmaxf(red[3], red[4]));
    for (int s = tid; s < S; s += 256) {
      float ab = aw[(long long)t * Smax + s] + nxt[s];
      if (ab > CTC_NEG_INF) atomicAdd(&occ[lab[s]], __expf(ab - mx));
    }
    __syncthreads();
    for (int v = tid; v < V; v += 256) {
      float lp = logp(t, v);
      float g = __expf(lp);
      float o = occ[v];
      if (o > 0.f) g -= __expf(__logf(o) + mx + nll - lp);
      gr[(long long)t * V + v] = g;
    }
    __syncthreads();
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
}

// logits [N,T,V] (bf16 if !is_f32), targets: concatenated int32 labels, tgt_off/tgt_len/in_len: int32 [N].
// lse: [N*T] scratch, alpha_ws: [N*T*Smax] scratch, loss: [N], grad: [N,T,V] fp32.  Smax >= 2*max(tgt_len)+1.
extern "C" int dlamd_ctc_loss(const void* logits, int is_f32, const void* targets, const void* tgt_off,
                              const void* tgt_len, const void* in_len, void* lse, void* alpha_ws, void* loss,
                              void* grad, int N, int T, int V, int Smax, int blank, int zero_infinity,
                              hipStream_t stream) {
  if (N <= 0 || T <= 0 || V <= 0) return 0;
  if (Smax < 1 || blank < 0 || blank >= V) return -1;
  size_t smem = (size_t)(3 * Smax + V + 8) * sizeof(float);
  if (smem > 150 * 1024) return -2;
  long long rows = (long long)N * T;
  unsigned g1 = (unsigned)((rows + 3) / 4);
  if (is_f32) {
    hipLaunchKernelGGL(ctc_lse_kernel<float>, dim3(g1), dim3(256), 0, stream, (const float*)logits, (float*)lse, rows, V);
    auto k = ctc_alpha_beta_kernel<float>;
    if (smem > 65536 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return -3;
    hipLaunchKernelGGL(k, dim3(N), dim3(256), smem, stream, (const float*)logits, (const float*)lse,
                       (const int*)targets, (const int*)tgt_off, (const int*)tgt_len, (const int*)in_len,
                       (float*)alpha_ws, (float*)loss, (float*)grad, T, V, Smax, blank, zero_infinity);
  } else {
    hipLaunchKernelGGL(ctc_lse_kernel<bf16_t>, dim3(g1), dim3(256), 0, stream, (const bf16_t*)logits, (float*)lse, rows, V);
    auto k = ctc_alpha_beta_kernel<bf16_t>;
    if (smem > 65536 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return -3;
    hipLaunchKernelGGL(k, dim3(N), dim3(256), smem, stream, (const bf16_t*)logits, (const float*)lse,
                       (const int*)targets, (const int*)tgt_off, (const int*)tgt_len, (const int*)in_len,
                       (float*)alpha_ws, (float*)loss, (float*)grad, T, V, Smax, blank, zero_infinity);
  }
  return DLAMD_CHECK_LAUNCH();
}
