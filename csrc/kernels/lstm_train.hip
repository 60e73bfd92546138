// Training kernels for the CRNN's bidirectional LSTM.
//
// PyTorch's LSTM under bf16 autocast on ROCm runs one GEMM + one cell kernel per time step, per
// direction, per layer, forward and backward: ~10 000 launches per training step (profiles/15), which
// is launch bound.  These two kernels run each recurrence as ONE persistent launch, like the inference
// kernel in lstm.hip (same operand orientation, same permuted gate-column order, same fragment-ordered
// W_hh streamed from L2), and exchange exactly what the backward pass needs:
//
//   forward : y[N,T,ndir*H] bf16, plus the post-activation gates (i,f,g,o; bf16, same layout as the
//             input projection) and the cell state c_t (fp32) of every step.
//   backward: walks the steps in reverse processing order.  A lane owns 4 hidden units x MT tiles of one
//             sequence: it adds the incoming dY to the recurrent dh (which the previous step's MFMA left in
//             its own accumulator registers), forms the four pre-activation gate gradients, writes them to
//             global memory (dG = gradient w.r.t. the input projection, and the operand of the weight-gradient
//             GEMMs that follow as ordinary library GEMMs) and to LDS, and the workgroup multiplies
//             dG_t [16 seq x 4H] by W_hh [4H x H] on MFMA to get dh for the next step.  dc stays in registers.
//
// One workgroup = 16 sequences of one direction, 4 waves; no grid-level synchronisation.
// Packed-sequence semantics (nn.utils.rnn.pack_padded_sequence): the reverse direction starts at len-1,
// steps at t >= len produce zeros and receive no gradient.
#include <stdlib.h>

#include "common.h"

__device__ __forceinline__ float lt_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float lt_tanh(float x) {
  float e = __expf(-2.f * fabsf(x));
  float t = (1.f - e) * __frcp_rn(1.f + e);
  return copysignf(t, x);
}

__device__ __forceinline__ uint32_t lt_pack2(float a, float b) {
  bf16x2 v = {(bf16_t)a, (bf16_t)b};
  return __builtin_bit_cast(uint32_t, v);
}

template <int H>
__global__ __launch_bounds__(256) void lstm_train_fwd_kernel(const bf16_t* __restrict__ xproj,   // [N,T,ndir*4H] permuted cols
                                                             const bf16_t* __restrict__ wfrag,   // [ndir][4H/16][H/32][64][8]
                                                             const int* __restrict__ lens,       // [N] or nullptr
                                                             bf16_t* __restrict__ y,             // [N,T,ndir*H]
                                                             bf16_t* __restrict__ gact,          // [N,T,ndir*4H]
                                                             float* __restrict__ cbuf,           // [N,T,ndir*H]
                                                             int N, int T, int ndir) {
  constexpr int G = 4 * H;
  constexpr int TILES = G / 16 / 4;
  constexpr int KS = H / 32;
  constexpr int HROW = H * 2 + 16;
  __shared__ __attribute__((aligned(16))) char hbuf[2][16 * HROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq_l = lane & 15, fq = lane >> 4;
  const int dir = blockIdx.y;
  const int n = blockIdx.x * 16 + seq_l;
  const bool seq_ok = n < N;
  const int len = seq_ok ? (lens ? min(max(lens[n], 0), T) : T) : 0;
  int maxlen = len;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));

  for (int i = tid; i < 2 * 16 * HROW / 4; i += 256) reinterpret_cast<uint32_t*>(&hbuf[0][0])[i] = 0u;
  __syncthreads();

  const bf16_t* wf = wfrag + (long long)dir * G * H + (long long)wave * TILES * KS * 64 * 8 + lane * 8;
  const long long xrow = (long long)ndir * G, yrow = (long long)ndir * H;
  const long long row0 = (long long)(seq_ok ? n : 0) * T;
  const long long gcol = (long long)dir * G + (wave * TILES) * 16 + fq * 4;
  const bf16_t* xp = xproj + row0 * xrow + gcol;
  bf16_t* gp = gact + row0 * xrow + gcol;
  bf16_t* yp = y + row0 * yrow + (long long)dir * H;
  float* cp = cbuf + row0 * yrow + (long long)dir * H;

  float cst[TILES];
#pragma unroll
  for (int i = 0; i < TILES; ++i) cst[i] = 0.f;

  for (int step = 0; step < maxlen; ++step) {
    const int cur = step & 1;
    const bool act = step < len;
    const bf16_t* wfs = wf;
    asm volatile("" : "+v"(wfs));   // keep the W_hh fragment loads inside the step (LICM would hoist and spill them)
    const int t = dir == 0 ? step : (len - 1 - step);
    f32x4 acc[TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      u32x2 raw = {0u, 0u};
      if (act) raw = *reinterpret_cast<const u32x2*>(xp + (long long)t * xrow + i * 16);
      acc[i] = f32x4{bf16_bits_to_f32(raw[0] & 0xffffu), bf16_bits_to_f32(raw[0] >> 16),
                     bf16_bits_to_f32(raw[1] & 0xffffu), bf16_bits_to_f32(raw[1] >> 16)};
    }
    bf16x8 hf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      hf[ks] = *reinterpret_cast<const bf16x8*>(&hbuf[cur][seq_l * HROW + ks * 64 + fq * 16]);
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      bf16x8 w[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) w[ks] = *reinterpret_cast<const bf16x8*>(wfs + (long long)(i * KS + ks) * 64 * 8);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks], hf[ks], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < TILES / 4; ++a) {
      bf16x4 hv;
      f32x4 cv;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = a * 4 + q;
        const float ig = lt_sigmoid(acc[i][0]), fg = lt_sigmoid(acc[i][1]), gg = lt_tanh(acc[i][2]), og = lt_sigmoid(acc[i][3]);
        const float c = fg * cst[i] + ig * gg;
        const float h = og * lt_tanh(c);
        if (act) {
          cst[i] = c;
          *reinterpret_cast<u32x2*>(gp + (long long)t * xrow + i * 16) = u32x2{lt_pack2(ig, fg), lt_pack2(gg, og)};
        }
        hv[q] = (bf16_t)h;
        cv[q] = c;
      }
      const int unit = (wave * (TILES / 4) + a) * 16 + fq * 4;
      if (act) {
        *reinterpret_cast<bf16x4*>(&hbuf[cur ^ 1][seq_l * HROW + unit * 2]) = hv;
        *reinterpret_cast<bf16x4*>(yp + (long long)t * yrow + unit) = hv;
        *reinterpret_cast<f32x4*>(cp + (long long)t * yrow + unit) = cv;
      } else {
        *reinterpret_cast<u32x2*>(&hbuf[cur ^ 1][seq_l * HROW + unit * 2]) =
            *reinterpret_cast<const u32x2*>(&hbuf[cur][seq_l * HROW + unit * 2]);
      }
    }
    __syncthreads();
  }
  if (seq_ok) {
    for (int t = len; t < T; ++t)
#pragma unroll
      for (int a = 0; a < TILES / 4; ++a) {
        const int unit = (wave * (TILES / 4) + a) * 16 + fq * 4;
        *reinterpret_cast<u32x2*>(yp + (long long)t * yrow + unit) = u32x2{0u, 0u};
      }
  }
}

template <int H>
__global__ __launch_bounds__(256) void lstm_train_bwd_kernel(const float* __restrict__ dy,       // [N,T,ndir*H]
                                                             const bf16_t* __restrict__ gact,    // [N,T,ndir*4H]
                                                             const float* __restrict__ cbuf,     // [N,T,ndir*H]
                                                             const bf16_t* __restrict__ wtfrag,  // [ndir][H/16][4H/32][64][8]
                                                             const int* __restrict__ lens,
                                                             bf16_t* __restrict__ dg,            // [N,T,ndir*4H], pre-zeroed
                                                             int N, int T, int ndir) {
  constexpr int G = 4 * H;
  constexpr int MT = H / 16 / 4;     // output (hidden-unit) tiles per wave
  constexpr int KS = G / 32;         // k-steps over the gate columns
  constexpr int GROW = G * 2 + 16;   // padded LDS row (bytes)
  __shared__ __attribute__((aligned(16))) char gbuf[16 * GROW];   // one buffer (33 KB at H=256), two barriers per step

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq_l = lane & 15, fq = lane >> 4;
  const int dir = blockIdx.y;
  const int n = blockIdx.x * 16 + seq_l;
  const bool seq_ok = n < N;
  const int len = seq_ok ? (lens ? min(max(lens[n], 0), T) : T) : 0;
  int maxlen = len;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));

  const bf16_t* wt = wtfrag + (long long)dir * G * H + (long long)wave * MT * KS * 64 * 8 + lane * 8;
  const long long xrow = (long long)ndir * G, yrow = (long long)ndir * H;
  const long long row0 = (long long)(seq_ok ? n : 0) * T;
  const bf16_t* gp = gact + row0 * xrow + (long long)dir * G + fq * 4;
  bf16_t* dgp = dg + row0 * xrow + (long long)dir * G + fq * 4;
  const float* cp = cbuf + row0 * yrow + (long long)dir * H + fq * 4;
  const float* dyp = dy + row0 * yrow + (long long)dir * H + fq * 4;

  f32x4 dh[MT];      // recurrent dh for units A*16 + fq*4 + j (the MFMA output of the previous step)
  float dcr[MT * 4]; // recurrent dc
#pragma unroll
  for (int a = 0; a < MT; ++a) dh[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MT * 4; ++i) dcr[i] = 0.f;

  // inputs of a step (dY, c_t, c_{t-1}, the four gates of every unit this lane owns): fetched one step ahead, under the
  // previous step's MFMAs - at the top of the step their latency (~2 us) was exposed in every one of the T steps
  f32x4 pdy[MT], pcv[MT], pcp[MT];
  u32x2 pg[MT][4];
  auto fetch = [&](int step) {
    const bool act = step >= 0 && step < len;
    const int t = dir == 0 ? step : (len - 1 - step);
    const int tp = dir == 0 ? t - 1 : t + 1;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int A = wave * MT + a;
      pdy[a] = f32x4{0.f, 0.f, 0.f, 0.f};
      pcv[a] = pdy[a];
      pcp[a] = pdy[a];
#pragma unroll
      for (int j = 0; j < 4; ++j) pg[a][j] = u32x2{0u, 0u};
      if (act) {
        pdy[a] = *reinterpret_cast<const f32x4*>(dyp + (long long)t * yrow + A * 16);
        pcv[a] = *reinterpret_cast<const f32x4*>(cp + (long long)t * yrow + A * 16);
        if (step > 0) pcp[a] = *reinterpret_cast<const f32x4*>(cp + (long long)tp * yrow + A * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) pg[a][j] = *reinterpret_cast<const u32x2*>(gp + (long long)t * xrow + (A * 4 + j) * 16);
      }
    }
  };
  fetch(maxlen - 1);

  for (int step = maxlen - 1; step >= 0; --step) {
    const bool act = step < len;
    const int t = dir == 0 ? step : (len - 1 - step);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int A = wave * MT + a;
      const f32x4 dyv = pdy[a], cv = pcv[a], cpv = pcp[a];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x2 graw = pg[a][j];
        const float ig = bf16_bits_to_f32(graw[0] & 0xffffu), fg = bf16_bits_to_f32(graw[0] >> 16);
        const float gg = bf16_bits_to_f32(graw[1] & 0xffffu), og = bf16_bits_to_f32(graw[1] >> 16);
        const float dht = dyv[j] + dh[a][j];
        const float tc = lt_tanh(cv[j]);
        const float dc = dcr[a * 4 + j] + dht * og * (1.f - tc * tc);
        const float d_o = dht * tc * og * (1.f - og);
        const float d_i = dc * gg * ig * (1.f - ig);
        const float d_g = dc * ig * (1.f - gg * gg);
        const float d_f = dc * cpv[j] * fg * (1.f - fg);
        u32x2 out = {0u, 0u};
        if (act) {
          dcr[a * 4 + j] = dc * fg;
          out = u32x2{lt_pack2(d_i, d_f), lt_pack2(d_g, d_o)};
          *reinterpret_cast<u32x2*>(dgp + (long long)t * xrow + (A * 4 + j) * 16) = out;
        }
        *reinterpret_cast<u32x2*>(&gbuf[seq_l * GROW + ((A * 4 + j) * 16 + fq * 4) * 2]) = out;
      }
    }
    __syncthreads();
    if (step == 0) break;   // dh of the first processed step has no consumer
    fetch(step - 1);        // in flight under the MFMAs below
    const bf16_t* wts = wt;
    asm volatile("" : "+v"(wts));
#pragma unroll
    for (int a = 0; a < MT; ++a) dh[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(&gbuf[seq_l * GROW + ks * 64 + fq * 16]);
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wts + (long long)(a * KS + ks) * 64 * 8);
        dh[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b, dh[a], 0, 0, 0);
      }
    }
    __syncthreads();   // every wave has read this step's dG tile before the next step overwrites it
  }
}

// Backward recurrence for H = 256 with most of W_hh^T resident: 8 waves, each owns 2 of the 16 hidden-unit tiles
// (64 fragments of 1 KB): K-steps 0 .. KR-1 live in registers, the next KL in LDS, the last NS = 32 - KR - KL are
// streamed from L2 in two batches through one small register buffer (the first batch is fetched during the previous
// step, the second while the resident K-steps are multiplied).  The streaming kernel above pulls all 512 KB per step
// through one CU's L2 port (~100 GB/s): 9 us per step.
template <int KR, int KL>
__global__ __launch_bounds__(512) void lstm_train_bwd_res_kernel(const float* __restrict__ dy, const bf16_t* __restrict__ gact,
                                                                 const float* __restrict__ cbuf, const bf16_t* __restrict__ wtfrag,
                                                                 const int* __restrict__ lens, bf16_t* __restrict__ dg, int N,
                                                                 int T, int ndir) {
  constexpr int H = 256, G = 4 * H, NW = 8;
  constexpr int MT = H / 16 / NW;    // 2 output tiles per wave
  constexpr int KS = G / 32;         // 32 k-steps over the gate columns
  constexpr int GROW = G * 2 + 16;
  constexpr int S0 = KR + KL;        // first streamed k-step
  constexpr int NS = KS - S0;
  constexpr int B1 = (NS + 1) / 2, B2 = NS - B1;   // k-steps of the two streamed batches
  static_assert(NS >= 2 && B1 >= B2, "streamed k-steps");
  extern __shared__ __attribute__((aligned(16))) char lt_smem[];
  char* gbuf = lt_smem;                         // [16][GROW]
  char* wl = lt_smem + 16 * GROW;               // [wave][KL][MT][64 lanes] x 16 B

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq_l = lane & 15, fq = lane >> 4;
  const int dir = blockIdx.y;
  const int n = blockIdx.x * 16 + seq_l;
  const bool seq_ok = n < N;
  const int len = seq_ok ? (lens ? min(max(lens[n], 0), T) : T) : 0;
  int maxlen = len;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));

  const bf16_t* wt = wtfrag + (long long)dir * G * H + (long long)wave * MT * KS * 64 * 8 + lane * 8;
  const long long xrow = (long long)ndir * G, yrow = (long long)ndir * H;
  const long long row0 = (long long)(seq_ok ? n : 0) * T;
  const bf16_t* gp = gact + row0 * xrow + (long long)dir * G + fq * 4;
  bf16_t* dgp = dg + row0 * xrow + (long long)dir * G + fq * 4;
  const float* cp = cbuf + row0 * yrow + (long long)dir * H + fq * 4;
  const float* dyp = dy + row0 * yrow + (long long)dir * H + fq * 4;

  bf16x8 wr[KR][MT];
#pragma unroll
  for (int k = 0; k < KR; ++k)
#pragma unroll
    for (int a = 0; a < MT; ++a) wr[k][a] = *reinterpret_cast<const bf16x8*>(wt + (long long)(a * KS + k) * 64 * 8);
  char* wlw = wl + ((wave * KL * MT) * 64 + lane) * 16;
#pragma unroll
  for (int k = 0; k < KL; ++k)
#pragma unroll
    for (int a = 0; a < MT; ++a)
      *reinterpret_cast<bf16x8*>(wlw + (k * MT + a) * 1024) =
          *reinterpret_cast<const bf16x8*>(wt + (long long)(a * KS + KR + k) * 64 * 8);

  f32x4 dh[MT];
  float dcr[MT * 4];
#pragma unroll
  for (int a = 0; a < MT; ++a) dh[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < MT * 4; ++i) dcr[i] = 0.f;

  f32x4 pdy[MT], pcv[MT], pcp[MT];
  u32x2 pg[MT][4];
  auto fetch = [&](int step) {
    const bool act = step >= 0 && step < len;
    const int t = dir == 0 ? step : (len - 1 - step);
    const int tp = dir == 0 ? t - 1 : t + 1;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int A = wave * MT + a;
      pdy[a] = f32x4{0.f, 0.f, 0.f, 0.f};
      pcv[a] = pdy[a];
      pcp[a] = pdy[a];
#pragma unroll
      for (int j = 0; j < 4; ++j) pg[a][j] = u32x2{0u, 0u};
      if (act) {
        pdy[a] = *reinterpret_cast<const f32x4*>(dyp + (long long)t * yrow + A * 16);
        pcv[a] = *reinterpret_cast<const f32x4*>(cp + (long long)t * yrow + A * 16);
        if (step > 0) pcp[a] = *reinterpret_cast<const f32x4*>(cp + (long long)tp * yrow + A * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) pg[a][j] = *reinterpret_cast<const u32x2*>(gp + (long long)t * xrow + (A * 4 + j) * 16);
      }
    }
  };
  fetch(maxlen - 1);
  bf16x8 ws[B1][MT];     // streamed fragments: batch 1 (k-steps S0 .. S0+B1-1), then batch 2 in the same registers
#pragma unroll
  for (int k = 0; k < B1; ++k)
#pragma unroll
    for (int a = 0; a < MT; ++a) ws[k][a] = *reinterpret_cast<const bf16x8*>(wt + (long long)(a * KS + S0 + k) * 64 * 8);
  __syncthreads();       // LDS-resident weights are in place

  for (int step = maxlen - 1; step >= 0; --step) {
    const bool act = step < len;
    const int t = dir == 0 ? step : (len - 1 - step);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int A = wave * MT + a;
      const f32x4 dyv = pdy[a], cv = pcv[a], cpv = pcp[a];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x2 graw = pg[a][j];
        const float ig = bf16_bits_to_f32(graw[0] & 0xffffu), fg = bf16_bits_to_f32(graw[0] >> 16);
        const float gg = bf16_bits_to_f32(graw[1] & 0xffffu), og = bf16_bits_to_f32(graw[1] >> 16);
        const float dht = dyv[j] + dh[a][j];
        const float tc = lt_tanh(cv[j]);
        const float dc = dcr[a * 4 + j] + dht * og * (1.f - tc * tc);
        const float d_o = dht * tc * og * (1.f - og);
        const float d_i = dc * gg * ig * (1.f - ig);
        const float d_g = dc * ig * (1.f - gg * gg);
        const float d_f = dc * cpv[j] * fg * (1.f - fg);
        u32x2 out = {0u, 0u};
        if (act) {
          dcr[a * 4 + j] = dc * fg;
          out = u32x2{lt_pack2(d_i, d_f), lt_pack2(d_g, d_o)};
          *reinterpret_cast<u32x2*>(dgp + (long long)t * xrow + (A * 4 + j) * 16) = out;
        }
        *reinterpret_cast<u32x2*>(&gbuf[seq_l * GROW + ((A * 4 + j) * 16 + fq * 4) * 2]) = out;
      }
    }
    __syncthreads();
    if (step == 0) break;
    fetch(step - 1);
    const bf16_t* wts = wt;
    asm volatile("" : "+v"(wts));          // keep the streamed loads inside the step loop
    const char* gb = gbuf + seq_l * GROW + fq * 16;
#pragma unroll
    for (int a = 0; a < MT; ++a) dh[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    // streamed batch 1 (arrived during the previous step), then batch 2 is requested into the same registers
#pragma unroll
    for (int k = 0; k < B1; ++k) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(gb + (S0 + k) * 64);
#pragma unroll
      for (int a = 0; a < MT; ++a) dh[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws[k][a], b, dh[a], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < B2; ++k)
#pragma unroll
      for (int a = 0; a < MT; ++a) ws[k][a] = *reinterpret_cast<const bf16x8*>(wts + (long long)(a * KS + S0 + B1 + k) * 64 * 8);
    // resident: registers
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(gb + k * 64);
#pragma unroll
      for (int a = 0; a < MT; ++a) dh[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[k][a], b, dh[a], 0, 0, 0);
    }
    // resident: LDS
#pragma unroll
    for (int k = 0; k < KL; ++k) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(gb + (KR + k) * 64);
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const bf16x8 w = *reinterpret_cast<const bf16x8*>(wlw + (k * MT + a) * 1024);
        dh[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b, dh[a], 0, 0, 0);
      }
    }
    // streamed batch 2
#pragma unroll
    for (int k = 0; k < B2; ++k) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(gb + (S0 + B1 + k) * 64);
#pragma unroll
      for (int a = 0; a < MT; ++a) dh[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws[k][a], b, dh[a], 0, 0, 0);
    }
    // batch 1 of the next step
#pragma unroll
    for (int k = 0; k < B1; ++k)
#pragma unroll
      for (int a = 0; a < MT; ++a) ws[k][a] = *reinterpret_cast<const bf16x8*>(wts + (long long)(a * KS + S0 + k) * 64 * 8);
    __syncthreads();   // every wave has read this step's dG tile before the next step overwrites it
  }
}

static int lt_launch_bwd_res(const void* dy, const void* gact, const void* cbuf, const void* wtfrag, const void* lens, void* dg, int N,
                             int T, int ndir, hipStream_t st) {
  constexpr int KR = 16, KL = 7;
  const size_t smem = 16 * (4 * 256 * 2 + 16) + (size_t)8 * KL * 2 * 1024;
  auto k = lstm_train_bwd_res_kernel<KR, KL>;
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return -5;
    set = true;
  }
  dim3 grid((N + 15) / 16, ndir);
  hipLaunchKernelGGL(k, grid, dim3(512), smem, st, (const float*)dy, (const bf16_t*)gact, (const float*)cbuf, (const bf16_t*)wtfrag,
                     (const int*)lens, (bf16_t*)dg, N, T, ndir);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

template <int H>
static int lt_launch_fwd(const void* xproj, const void* wfrag, const void* lens, void* y, void* gact, void* cbuf, int N, int T,
                         int ndir, hipStream_t st) {
  dim3 grid((N + 15) / 16, ndir), block(256);
  hipLaunchKernelGGL(lstm_train_fwd_kernel<H>, grid, block, 0, st, (const bf16_t*)xproj, (const bf16_t*)wfrag, (const int*)lens,
                     (bf16_t*)y, (bf16_t*)gact, (float*)cbuf, N, T, ndir);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

template <int H>
static int lt_launch_bwd(const void* dy, const void* gact, const void* cbuf, const void* wtfrag, const void* lens, void* dg, int N,
                         int T, int ndir, hipStream_t st) {
  dim3 grid((N + 15) / 16, ndir), block(256);
  hipLaunchKernelGGL(lstm_train_bwd_kernel<H>, grid, block, 0, st, (const float*)dy, (const bf16_t*)gact, (const float*)cbuf,
                     (const bf16_t*)wtfrag, (const int*)lens, (bf16_t*)dg, N, T, ndir);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

// Shapes are the caller's contract (ops/lstm_train.py checks tensor sizes against N, T, H, ndir).
// lstm.hip: the inference kernel with three quarters of W_hh resident (registers + LDS), saving gates and cell states
int dlamd_lstm4_train_fwd(const void* xproj, const void* wfrag, const void* lens, void* y, void* gact, void* cbuf, int N, int T,
                          int ndir, hipStream_t stream);

extern "C" int dlamd_lstm_train_fwd(const void* xproj, const void* wfrag, const void* lens, void* y, void* gact, void* cbuf, int N,
                                    int T, int H, int ndir, hipStream_t st) {
  if (!xproj || !wfrag || !y || !gact || !cbuf || N <= 0 || T <= 0 || ndir < 1 || ndir > 2) return -1;
  switch (H) {
    case 64: return lt_launch_fwd<64>(xproj, wfrag, lens, y, gact, cbuf, N, T, ndir, st);
    case 128: return lt_launch_fwd<128>(xproj, wfrag, lens, y, gact, cbuf, N, T, ndir, st);
    case 256: {
      // H = 256: the kernel that keeps three quarters of W_hh in registers + LDS (4.0 us per step against 7.0 with the
      // whole matrix streamed from L2 every step: one CU pulls ~100 GB/s); DLAMD_LSTM_TRAIN_STREAMED=1 keeps the
      // streaming kernel
      static const bool streamed = getenv("DLAMD_LSTM_TRAIN_STREAMED") != nullptr && atoi(getenv("DLAMD_LSTM_TRAIN_STREAMED")) != 0;
      if (!streamed) return dlamd_lstm4_train_fwd(xproj, wfrag, lens, y, gact, cbuf, N, T, ndir, st);
      return lt_launch_fwd<256>(xproj, wfrag, lens, y, gact, cbuf, N, T, ndir, st);
    }
    default: return -2;
  }
}

extern "C" int dlamd_lstm_train_bwd(const void* dy, const void* gact, const void* cbuf, const void* wtfrag, const void* lens,
                                    void* dg, int N, int T, int H, int ndir, hipStream_t st) {
  if (!dy || !gact || !cbuf || !wtfrag || !dg || N <= 0 || T <= 0 || ndir < 1 || ndir > 2) return -1;
  switch (H) {
    case 64: return lt_launch_bwd<64>(dy, gact, cbuf, wtfrag, lens, dg, N, T, ndir, st);
    case 128: return lt_launch_bwd<128>(dy, gact, cbuf, wtfrag, lens, dg, N, T, ndir, st);
    case 256: {
      static const bool streamed = getenv("DLAMD_LSTM_TRAIN_STREAMED") != nullptr && atoi(getenv("DLAMD_LSTM_TRAIN_STREAMED")) != 0;
      if (!streamed) return lt_launch_bwd_res(dy, gact, cbuf, wtfrag, lens, dg, N, T, ndir, st);
      return lt_launch_bwd<256>(dy, gact, cbuf, wtfrag, lens, dg, N, T, ndir, st);
    }
    default: return -2;
  }
}
