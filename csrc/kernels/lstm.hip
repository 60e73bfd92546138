// Bidirectional LSTM recurrence for the CRNN recognizer (hard part 7.4-4).
//
// The input projection x.W_ih^T + b for ALL time steps is one large MFMA GEMM
// (conv_igemm, 1x1 path).  This kernel runs only the sequential part and is
// persistent over time: one workgroup owns 16 sequences of one direction and
// loops over T internally, so there is no grid-level synchronisation at all:
//   * h_{t-1} lives in LDS (bf16, double buffered, rows padded to 528 B so the
//     16 sequence rows fall on distinct banks for ds_read_b128),
//   * c_t lives in registers (a lane owns 4 gates x 1 hidden unit x 1 sequence
//     per MFMA tile, so the cell update is lane-local),
//   * W_hh is streamed from L2 every step in MFMA *fragment order* (host
//     pre-swizzles it so that every wave-level load is one contiguous 1 KiB),
//     all workgroups read the same 2*H*4H bytes so it stays L2 resident.
// MFMA orientation: A = W_hh fragment (rows = gate columns), B = h fragment
// (cols = sequences): D[row][col] puts 4 consecutive gate rows of one sequence
// in one lane.  Gate rows are laid out as r = tile*16 + fq*4 + gate with
// unit(tile, fq) = (tile/4)*16 + fq*4 + tile%4, so four consecutive tiles give
// a lane four consecutive hidden units -> 8-byte packed h stores.
#include "common.h"

template <int H>
__global__ __launch_bounds__(256) void lstm_bidir_kernel(const bf16_t* __restrict__ xproj,   // [N, T, ndir*4H] permuted cols
                                                         const bf16_t* __restrict__ wfrag,   // [ndir][4H/16][H/32][64][8]
                                                         const int* __restrict__ lens,       // [N] or nullptr
                                                         bf16_t* __restrict__ y,             // [N, T, ndir*H]
                                                         int N, int T, int ndir) {
  constexpr int G = 4 * H;            // gate columns
  constexpr int TILES = G / 16 / 4;   // row tiles per wave
  constexpr int KS = H / 32;          // k-steps
  constexpr int HROW = H * 2 + 16;    // padded LDS row (bytes)
  __shared__ __attribute__((aligned(16))) char hbuf[2][16 * HROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int seq_l = lane & 15, fq = lane >> 4;
  const int dir = blockIdx.y;
  const int n = blockIdx.x * 16 + seq_l;
  const bool seq_ok = n < N;
  const int len = seq_ok ? (lens ? min(lens[n], T) : T) : 0;
  // longest sequence in this workgroup
  int maxlen = len;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));

  for (int i = tid; i < 2 * 16 * HROW / 4; i += 256) reinterpret_cast<uint32_t*>(&hbuf[0][0])[i] = 0u;
  __syncthreads();

  const bf16_t* wf = wfrag + (long long)dir * G * H + (long long)wave * TILES * KS * 64 * 8 + lane * 8;
  const long long xrow = (long long)ndir * G;       // elements per (n, t) row of xproj
  const long long yrow = (long long)ndir * H;
  const bf16_t* xp = xproj + (long long)(seq_ok ? n : 0) * T * xrow + (long long)dir * G + (wave * TILES) * 16 + fq * 4;
  bf16_t* yp = y + (long long)(seq_ok ? n : 0) * T * yrow + (long long)dir * H;

  float cst[TILES];
#pragma unroll
  for (int i = 0; i < TILES; ++i) cst[i] = 0.f;

  for (int step = 0; step < maxlen; ++step) {
    const int cur = step & 1;
    const bool act = step < len;
    const int t = dir == 0 ? step : (len - 1 - step);
    // gate pre-activations from the input projection
    f32x4 acc[TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      u32x2 raw = {0u, 0u};
      if (act) raw = *reinterpret_cast<const u32x2*>(xp + (long long)t * xrow + i * 16);
      acc[i] = f32x4{bf16_bits_to_f32(raw[0] & 0xffffu), bf16_bits_to_f32(raw[0] >> 16),
                     bf16_bits_to_f32(raw[1] & 0xffffu), bf16_bits_to_f32(raw[1] >> 16)};
    }
    // h_{t-1} fragments (shared by every row tile)
    bf16x8 hf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      hf[ks] = *reinterpret_cast<const bf16x8*>(&hbuf[cur][seq_l * HROW + ks * 64 + fq * 16]);
    // recurrent product, W_hh streamed from L2 in fragment order
#pragma unroll
    for (int i = 0; i < TILES; ++i) {
      bf16x8 w[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        w[ks] = *reinterpret_cast<const bf16x8*>(wf + (long long)(i * KS + ks) * 64 * 8);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ks], hf[ks], acc[i], 0, 0, 0);
    }
    // cell update; tiles 4a..4a+3 hold units a*16 + fq*4 + (0..3) of this wave's block
#pragma unroll
    for (int a = 0; a < TILES / 4; ++a) {
      bf16x4 hv;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = a * 4 + q;
        float ig = 1.f / (1.f + __expf(-acc[i][0]));
        float fg = 1.f / (1.f + __expf(-acc[i][1]));
        float gg = tanhf(acc[i][2]);
        float og = 1.f / (1.f + __expf(-acc[i][3]));
        float c = fg * cst[i] + ig * gg;
        float h = og * tanhf(c);
        if (act) cst[i] = c;
        hv[q] = (bf16_t)h;
      }
      const int unit = (wave * (TILES / 4) + a) * 16 + fq * 4;
      if (act) {
        *reinterpret_cast<bf16x4*>(&hbuf[cur ^ 1][seq_l * HROW + unit * 2]) = hv;
        *reinterpret_cast<bf16x4*>(yp + (long long)t * yrow + unit) = hv;
      } else {
        // frozen state: carry h forward unchanged
        *reinterpret_cast<u32x2*>(&hbuf[cur ^ 1][seq_l * HROW + unit * 2]) =
            *reinterpret_cast<const u32x2*>(&hbuf[cur][seq_l * HROW + unit * 2]);
      }
    }
    __syncthreads();
  }
  // zero the padded tail of y so downstream GEMMs see clean rows
  if (seq_ok) {
    for (int t = len; t < T; ++t)
#pragma unroll
      for (int a = 0; a < TILES / 4; ++a) {
        const int unit = (wave * (TILES / 4) + a) * 16 + fq * 4;
        *reinterpret_cast<u32x2*>(yp + (long long)t * yrow + unit) = u32x2{0u, 0u};
      }
  }
}

// xproj: [N, T, ndir*4H] bf16 with gate columns in the permuted order (see
// ops/lstm.py: lstm_gate_permutation); wfrag: fragment-ordered W_hh per direction.
extern "C" int dlamd_lstm_bidir(const void* xproj, const void* wfrag, const void* lens, void* y, int N, int T, int H,
                                int ndir, hipStream_t stream) {
  if (N <= 0 || T <= 0) return 0;
  if (ndir < 1 || ndir > 2) return -1;
  dim3 grid((N + 15) / 16, ndir);
  if (H == 256)
    hipLaunchKernelGGL(lstm_bidir_kernel<256>, grid, dim3(256), 0, stream, (const bf16_t*)xproj, (const bf16_t*)wfrag,
                       (const int*)lens, (bf16_t*)y, N, T, ndir);
  else if (H == 128)
    hipLaunchKernelGGL(lstm_bidir_kernel<128>, grid, dim3(256), 0, stream, (const bf16_t*)xproj, (const bf16_t*)wfrag,
                       (const int*)lens, (bf16_t*)y, N, T, ndir);
  else if (H == 64)
    hipLaunchKernelGGL(lstm_bidir_kernel<64>, grid, dim3(256), 0, stream, (const bf16_t*)xproj, (const bf16_t*)wfrag,
                       (const int*)lens, (bf16_t*)y, N, T, ndir);
  else
    return -2;
  return DLAMD_CHECK_LAUNCH();
}
