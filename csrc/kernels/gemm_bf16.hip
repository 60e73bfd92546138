// Persistent, software-pipelined bf16 GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   y[M, N] = act(x[M, K] . w[N, K]^T + bias (+ res))          (GEMM: x rows are K-contiguous)
//   y[pixel, Cout] = the same with the row gathered per filter tap (conv: NHWC, Cin % 64 == 0)
//
// This is the kernel behind every plain linear layer of the recognizers (LSTM input projections, ViT QKV /
// proj / fc1 / fc2, decoder projections) and behind the wide 3x3 convolutions of the detector and the CRNN
// (the reference runs those layers inside magic-pdf's CPU torch models, services/ocr_service/ocr_service.py:215-220).
//
// Structure (one 512-thread workgroup per CU, 256 x 256 x 64 tiles, 128 KiB of LDS):
//   * 8 waves = 2 (wr) x 4 (wc).  The tile is cut in two row halves A0/A1 (128 rows each) and two column halves
//     B0/B1.  Wave (wr, wc) owns rows wr*64..+63 of BOTH row halves and columns wc*32..+31 of BOTH column halves,
//     so that every wave consumes the four half-tiles in the same order and one K-tile is four phases:
//         phase 1: (A0 x B0)   reads A0, B0 fragments        phase 3: (A1 x B1)   reads A1 fragments
//         phase 2: (A0 x B1)   reads B1 fragments            phase 4: (A1 x B0)   no LDS reads
//     16 MFMA 16x16x32 per wave and phase.
//   * Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4), never through a VGPR.  One half-tile
//     (2 DMA instructions per thread) is issued per phase, into the half-tile slot whose last fragment read lies
//     two phases back; a half-tile is therefore in flight for five phases before its first read.  The only wait in
//     the loop is a counted s_waitcnt vmcnt(8) per phase (four half-tiles stay in flight across every barrier).
//   * LDS rows are 128 B (64 bf16); the 16-byte chunk c of row r is stored at chunk c ^ (r & 7), which makes the
//     ds_read_b128 fragment reads conflict-free.  An LDS-DMA writes lane-linear, so the XOR is applied to the
//     per-lane SOURCE address.
//   * The two wave rows run half a phase apart (wr = 1 passes one extra barrier at the start, wr = 0 one at the
//     end): while one group issues its MFMA cluster the other reads fragments and issues DMA, so the two waves
//     that share a SIMD alternate on its matrix pipe instead of colliding on it.
//   * Persistent over output tiles: the DMA stream simply continues into the next tile, and the epilogue of a
//     quadrant (bias from an LDS copy, residual, activation, bf16 pack, v_permlane16_swap so that a lane owns
//     8 consecutive channels, one 16-byte store) runs in the NEXT phase's load section, under the other group's
//     MFMA cluster.  Tiles are dealt so that each XCD (private L2) walks consecutive tiles, which share x rows.
//   * MFMA operands are swapped (weights = A operand, x rows = B operand): a lane's four accumulator values are
//     four consecutive output channels of one row.
//
// Synchronisation argument, per half-tile slot (G0 = waves wr 0, G1 = waves wr 1, G1 half a phase behind):
//   RAW  a fragment read in phase p needs every wave's DMA into the slot complete: each wave's counted vmcnt in
//        phase p-1 retires it (it is older than the 8 youngest operations there), and both groups' phase p-1 load
//        sections end at a barrier every wave passes before any wave starts the load section of phase p.
//   WAR  a DMA issued in phase p+2 overwrites a slot last read in phase p: the reads are retired by the
//        s_waitcnt lgkmcnt(0) that opens the MFMA section of phase p, and G1's MFMA section of phase p ends at the
//        barrier that opens G0's load section of phase p+2 (for G1 itself one barrier later).
#include <stdlib.h>

#include "common.h"

struct G8Args {
  const bf16_t* x;
  const bf16_t* w;
  const float* bias;
  const bf16_t* res;
  void* y;
  int M, N, K, nkt;
  int x_cs8;            // distance between consecutive x rows / pixels in 16-byte chunks
  int H, W, Cin, Ho, Wo, kh, kw, sh, sw, ph, pw;   // conv geometry (CONV instantiation only)
  int act, res_mode, res_cstride, y_cstride, y_coff, y_f32;
  int tiles_n, tiles;
};

namespace {

constexpr int G8_BUF = 65536;                 // one K-tile: A0 | B0 | B1 | A1, 16 KiB each
constexpr int G8_A0 = 0, G8_B0 = 16384, G8_B1 = 32768, G8_A1 = 49152;
constexpr int G8_BIAS = 2 * G8_BUF;           // two 1 KiB bias images (alternating output tiles)
constexpr int G8_LDS = G8_BIAS + 2048;
constexpr int G8_INVALID = 0x40000000;

__device__ __attribute__((aligned(256))) uint8_t g8_zero_page[256];

template <int OFF>
__device__ __forceinline__ bf16x8 g8_lds_read(const unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

__device__ __forceinline__ void g8_dma16(const bf16_t* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// DMA stream position: which K-tile of which output tile the next half-tile comes from, plus this thread's two
// source rows for each of the two half-tiles the cursor serves (cursor F: A0 + B0, cursor S: B1 + A1).
template <bool CONV>
struct G8Cursor {
  int tile, kt, flat;
  int dy, dx, cin0;
  int a_base[2], a_hw[2], b_base[2];
};

template <bool CONV, int AH, int BH>
__device__ __forceinline__ void g8_cursor_rows(G8Cursor<CONV>& c, const G8Args& p, const int rp, const int ch) {
  const bool live = c.tile < p.tiles;
  const int tile = live ? c.tile : 0;
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = tm * 256 + AH * 128 + j * 64 + rp;
    const int n = tn * 256 + BH * 128 + j * 64 + rp;
    c.b_base[j] = (live && n < p.N) ? n * (p.K >> 3) + ch : G8_INVALID;
    if (!CONV) {
      c.a_base[j] = (live && m < p.M) ? m * p.x_cs8 + ch : G8_INVALID;
      c.a_hw[j] = 0;
    } else {
      const int hw = p.Ho * p.Wo;
      const int mm = (live && m < p.M) ? m : 0;
      const int b = mm / hw, rem = mm - b * hw;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int hi0 = ho * p.sh - p.ph, wi0 = wo * p.sw - p.pw;
      c.a_base[j] = (live && m < p.M) ? ((b * p.H + hi0) * p.W + wi0) * p.x_cs8 + ch : G8_INVALID;
      c.a_hw[j] = (hi0 << 16) | (wi0 & 0xffff);
    }
  }
}

template <bool CONV>
__device__ __forceinline__ const bf16_t* g8_src_a(const G8Cursor<CONV>& c, const G8Args& p, const int j, const bf16_t* zero) {
  int off;
  bool ok = c.a_base[j] != G8_INVALID;
  if (!CONV) {
    off = c.a_base[j] + c.kt * 8;
  } else {
    const int hi = (c.a_hw[j] >> 16) + c.dy, wi = (int)(short)(c.a_hw[j] & 0xffff) + c.dx;
    ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    off = c.a_base[j] + (c.dy * p.W + c.dx) * p.x_cs8 + (c.cin0 >> 3);
  }
  return ok ? p.x + (long long)off * 8 : zero;
}

template <bool CONV>
__device__ __forceinline__ const bf16_t* g8_src_b(const G8Cursor<CONV>& c, const G8Args& p, const int j, const bf16_t* zero) {
  return c.b_base[j] != G8_INVALID ? p.w + (long long)(c.b_base[j] + c.kt * 8) * 8 : zero;
}

template <bool CONV, int AH, int BH>
__device__ __forceinline__ void g8_cursor_advance(G8Cursor<CONV>& c, const G8Args& p, const int stride, const int rp, const int ch) {
  ++c.kt;
  ++c.flat;
  if (CONV) {
    c.cin0 += 64;
    if (c.cin0 == p.Cin) {
      c.cin0 = 0;
      if (++c.dx == p.kw) { c.dx = 0; ++c.dy; }
    }
  }
  if (c.kt == p.nkt) {
    c.kt = 0;
    c.dy = c.dx = c.cin0 = 0;
    c.tile = c.tile < p.tiles ? c.tile + stride : c.tile;
    g8_cursor_rows<CONV, AH, BH>(c, p, rp, ch);
  }
}

// erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below bf16 resolution): a dozen instructions where the
// library erff is several dozen, and this epilogue is instantiated five times per kernel.
__device__ __forceinline__ float g8_gelu(const float v) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __frcp_rn(1.f + 0.3275911f * x);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = 1.f - poly * __expf(-x * x);
  return 0.5f * v * (1.f + (v < 0.f ? -e : e));
}

// v_permlane16_swap_b32: lanes 16-31 / 48-63 of the first operand change places with lanes 0-15 / 32-47 of the second.
// Inline asm on purpose: hipcc (ROCm 7.2) folds several calls of __builtin_amdgcn_permlane16_swap in one unrolled loop
// into a single swap whose first result is used for every element.  The s_nop covers the VALU-write -> permlane-read
// hazard, which the compiler does not pad inside an asm statement.
__device__ __forceinline__ void g8_swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// Epilogue of one quadrant (row half AH x column half BH) of the tile at (m0, n0); clears its accumulators.
// The two 16-column accumulator tiles of a row are first traded between half-rows of lanes (v_permlane16_swap:
// lanes 16-31 / 48-63 of the first operand change places with lanes 0-15 / 32-47 of the second), after which a lane
// owns 8 CONSECUTIVE channels of its row - fq 0 -> 0..7, 1 -> 16..23, 2 -> 8..15, 3 -> 24..31 - so bias, residual
// and the store are 16- or 32-byte accesses.
template <int AH, int BH, bool RES>
__device__ __forceinline__ void g8_epilogue(const G8Args& p, f32x4 (&acc)[2][2][4][2], const int m0, const int n0,
                                            const char* bias_img, const int wr, const int wc, const int fr, const int fq) {
  const int nt = BH * 128 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;     // first of this lane's 8 columns, inside the tile
  const int n = n0 + nt;
  const bool col_ok = n < p.N;
  f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if (p.bias) {
    b0 = *reinterpret_cast<const f32x4*>(bias_img + nt * 4);
    b1 = *reinterpret_cast<const f32x4*>(bias_img + nt * 4 + 16);
  }
  const int mrow = m0 + AH * 128 + wr * 64 + fr;
  u32x4 rr[4];
  if (RES) {
    const int nc = col_ok ? n : 0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = mrow + mi * 16;
      rr[mi] = *reinterpret_cast<const u32x4*>(p.res + (long long)(m < p.M ? m : p.M - 1) * p.res_cstride + nc);
    }
  }
  const float lo = p.act == ACT_RELU ? 0.f : -3.0e38f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = mrow + mi * 16;
    float v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float lo4 = acc[AH][BH][mi][0][q], hi4 = acc[AH][BH][mi][1][q];
      g8_swap16(lo4, hi4);
      v[q] = lo4 + b0[q];
      v[4 + q] = hi4 + b1[q];
    }
    acc[AH][BH][mi][0] = acc[AH][BH][mi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (RES) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[2 * q] += bf16_bits_to_f32(rr[mi][q] & 0xffffu);
        v[2 * q + 1] += bf16_bits_to_f32(rr[mi][q] >> 16);
      }
    }
    if (p.act == ACT_GELU) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = g8_gelu(v[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], lo);
    }
    if (m < p.M && col_ok) {
      if (p.y_f32) {
        float* yp = reinterpret_cast<float*>(p.y) + (long long)m * p.y_cstride + p.y_coff + n;
        *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(yp + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        const bf16x8 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3],
                          (bf16_t)v[4], (bf16_t)v[5], (bf16_t)v[6], (bf16_t)v[7]};
        *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.y) + (long long)m * p.y_cstride + p.y_coff + n) = o;
      }
    }
  }
}

#define G8_PHASE_SYNC()                                   \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1);

#define G8_PHASE_END()                                    \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);

#define G8_MFMA_QUAD(AH, BH, WF)                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                            \
  _Pragma("unroll") for (int nj = 0; nj < 2; ++nj)                                                            \
      acc[AH][BH][mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nj][ks], xf[mi][ks], acc[AH][BH][mi][nj], 0, 0, 0);

template <bool CONV, bool RES>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const G8Args p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int rp = tid >> 3;                       // DMA role: row rp of each 64-row pass ...
  const int ch = (tid & 7) ^ (rp & 7);           // ... and the logical 16-byte chunk whose swizzled slot is tid & 7
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g8_zero_page) + (tid & 7) * 8;

  const int G = gridDim.x;
  const int first_tile = dlamd_xcd_remap(blockIdx.x, G);
  const int my_tiles = first_tile < p.tiles ? (p.tiles - first_tile + G - 1) / G : 0;
  if (my_tiles == 0) return;

  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const unsigned fr_off = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const unsigned xa = lds0 + fr_off + wr * 8192;   // x fragments: row wr*64 + mi*16 + fr of a row half
  const unsigned wa = lds0 + fr_off + wc * 4096;   // w fragments: row wc*32 + nj*16 + fr of a column half
  char* dma_wave = smem + wid * 1024;

  G8Cursor<CONV> cf, cs;                           // F: A0 + B0 (two K-tiles ahead), S: B1 + A1 (one ahead)
  cf.tile = cs.tile = first_tile;
  cf.kt = cs.kt = cf.flat = cs.flat = 0;
  cf.dy = cf.dx = cf.cin0 = cs.dy = cs.dx = cs.cin0 = 0;
  g8_cursor_rows<CONV, 0, 0>(cf, p, rp, ch);
  g8_cursor_rows<CONV, 1, 1>(cs, p, rp, ch);

  auto bias_dma = [&](const int tile, const int which) {
    if (p.bias != nullptr && wid < 4) {
      const int tn = tile % p.tiles_n;
      int n = tn * 256 + wid * 64 + lane;
      n = n < p.N ? n : p.N - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + n),
                                       (__attribute__((address_space(3))) void*)(smem + G8_BIAS + which * 1024 + wid * 256), 4, 0, 0);
    }
  };
#define G8_STAGE_A(C, SLOT)                                                                             \
  {                                                                                                     \
    char* d = dma_wave + ((C.flat & 1) ? G8_BUF : 0) + SLOT;                                            \
    g8_dma16(g8_src_a<CONV>(C, p, 0, zero), d);                                                         \
    g8_dma16(g8_src_a<CONV>(C, p, 1, zero), d + 8192);                                                  \
  }
#define G8_STAGE_B(C, SLOT)                                                                             \
  {                                                                                                     \
    char* d = dma_wave + ((C.flat & 1) ? G8_BUF : 0) + SLOT;                                            \
    g8_dma16(g8_src_b<CONV>(C, p, 0, zero), d);                                                         \
    g8_dma16(g8_src_b<CONV>(C, p, 1, zero), d + 8192);                                                  \
  }

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: K-tile 0 completely, A0 + B0 of K-tile 1
  bias_dma(first_tile, 0);
  G8_STAGE_A(cf, G8_A0);
  G8_STAGE_B(cf, G8_B0);
  G8_STAGE_B(cs, G8_B1);
  G8_STAGE_A(cs, G8_A1);
  g8_cursor_advance<CONV, 1, 1>(cs, p, G, rp, ch);
  g8_cursor_advance<CONV, 0, 0>(cf, p, G, rp, ch);
  G8_STAGE_A(cf, G8_A0);
  G8_STAGE_B(cf, G8_B0);
  g8_cursor_advance<CONV, 0, 0>(cf, p, G, rp, ch);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // A0 and B0 of K-tile 0 have landed
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();                  // G1 runs half a phase behind G0
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 xf[4][2], w0[2][2], w1[2][2];
  int tile = first_tile, kt = 0, tile_seq = 0;
  int m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * 256;
  int pm0 = 0, pn0 = 0;
  const int total = my_tiles * p.nkt;

  for (int T = 0; T < total; ++T) {
    const unsigned boff = (T & 1) ? G8_BUF : 0;
    const bool first = kt == 0, last = kt == p.nkt - 1;
    const unsigned xb = xa + boff, wb = wa + boff;
    const unsigned xb1 = xb ^ 64, wb1 = wb ^ 64;
    const char* bias_img = smem + G8_BIAS + (tile_seq & 1) * 1024;

    // ---------------- phase 1: A0 x B0
    if (first) {
      if (T > 0) {
        g8_epilogue<1, 0, RES>(p, acc, pm0, pn0, smem + G8_BIAS + ((tile_seq ^ 1) & 1) * 1024, wr, wc, fr, fq);
        bias_dma(tile, tile_seq & 1);
      }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if (mi == 0) { xf[0][0] = g8_lds_read<G8_A0 + 0 * 2048>(xb); xf[0][1] = g8_lds_read<G8_A0 + 0 * 2048>(xb1); }
      if (mi == 1) { xf[1][0] = g8_lds_read<G8_A0 + 1 * 2048>(xb); xf[1][1] = g8_lds_read<G8_A0 + 1 * 2048>(xb1); }
      if (mi == 2) { xf[2][0] = g8_lds_read<G8_A0 + 2 * 2048>(xb); xf[2][1] = g8_lds_read<G8_A0 + 2 * 2048>(xb1); }
      if (mi == 3) { xf[3][0] = g8_lds_read<G8_A0 + 3 * 2048>(xb); xf[3][1] = g8_lds_read<G8_A0 + 3 * 2048>(xb1); }
    }
    w0[0][0] = g8_lds_read<G8_B0>(wb); w0[0][1] = g8_lds_read<G8_B0>(wb1);
    w0[1][0] = g8_lds_read<G8_B0 + 2048>(wb); w0[1][1] = g8_lds_read<G8_B0 + 2048>(wb1);
    G8_STAGE_B(cs, G8_B1);
    G8_PHASE_SYNC();
    G8_MFMA_QUAD(0, 0, w0);
    G8_PHASE_END();

    // ---------------- phase 2: A0 x B1
    if (last) g8_epilogue<0, 0, RES>(p, acc, m0, n0, bias_img, wr, wc, fr, fq);
    w1[0][0] = g8_lds_read<G8_B1>(wb); w1[0][1] = g8_lds_read<G8_B1>(wb1);
    w1[1][0] = g8_lds_read<G8_B1 + 2048>(wb); w1[1][1] = g8_lds_read<G8_B1 + 2048>(wb1);
    G8_STAGE_A(cs, G8_A1);
    g8_cursor_advance<CONV, 1, 1>(cs, p, G, rp, ch);
    G8_PHASE_SYNC();
    G8_MFMA_QUAD(0, 1, w1);
    G8_PHASE_END();

    // ---------------- phase 3: A1 x B1
    if (last) g8_epilogue<0, 1, RES>(p, acc, m0, n0, bias_img, wr, wc, fr, fq);
    xf[0][0] = g8_lds_read<G8_A1 + 0 * 2048>(xb); xf[0][1] = g8_lds_read<G8_A1 + 0 * 2048>(xb1);
    xf[1][0] = g8_lds_read<G8_A1 + 1 * 2048>(xb); xf[1][1] = g8_lds_read<G8_A1 + 1 * 2048>(xb1);
    xf[2][0] = g8_lds_read<G8_A1 + 2 * 2048>(xb); xf[2][1] = g8_lds_read<G8_A1 + 2 * 2048>(xb1);
    xf[3][0] = g8_lds_read<G8_A1 + 3 * 2048>(xb); xf[3][1] = g8_lds_read<G8_A1 + 3 * 2048>(xb1);
    G8_STAGE_A(cf, G8_A0);
    G8_PHASE_SYNC();
    G8_MFMA_QUAD(1, 1, w1);
    G8_PHASE_END();

    // ---------------- phase 4: A1 x B0
    if (last) g8_epilogue<1, 1, RES>(p, acc, m0, n0, bias_img, wr, wc, fr, fq);
    G8_STAGE_B(cf, G8_B0);
    g8_cursor_advance<CONV, 0, 0>(cf, p, G, rp, ch);
    G8_PHASE_SYNC();
    G8_MFMA_QUAD(1, 0, w0);
    G8_PHASE_END();

    if (last) {
      pm0 = m0; pn0 = n0;
      kt = 0;
      tile += G;
      ++tile_seq;
      const int tcl = tile < p.tiles ? tile : 0;
      m0 = (tcl / p.tiles_n) * 256;
      n0 = (tcl % p.tiles_n) * 256;
    } else {
      ++kt;
    }
  }
  g8_epilogue<1, 0, RES>(p, acc, pm0, pn0, smem + G8_BIAS + ((tile_seq ^ 1) & 1) * 1024, wr, wc, fr, fq);
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 0) __builtin_amdgcn_s_barrier();                  // balances G1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no LDS-DMA may outlive the workgroup's LDS allocation
}

#undef G8_STAGE_A
#undef G8_STAGE_B

template <bool CONV, bool RES>
int g8_launch(const G8Args& a, hipStream_t stream) {
  auto kern = gemm8p_kernel<CONV, RES>;
  static bool set = false;
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS) != hipSuccess)
      return -2;
    set = true;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -2;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int grid = a.tiles < n_cu ? a.tiles : n_cu;
  if (const char* e = getenv("DLAMD_GEMM8_GRID")) {           // tests: several tiles per workgroup on small problems
    const int v = atoi(e);
    if (v > 0 && v < grid) grid = v;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), G8_LDS, stream, a);
  return DLAMD_CHECK_LAUNCH();
}

}  // namespace

// Whether dlamd_gemm8 serves a shape (host-side contract of the kernel; callers fall back to conv_igemm otherwise).
extern "C" int dlamd_gemm8_serves(int M, int N, int K, int Cin, int x_cstride, int kh, int kw, int act, int res_mode,
                                  int out_mode, int y_cstride, int y_coff) {
  if (M <= 0 || N < 8 || K < 128 || K % 64 != 0 || Cin % 64 != 0 || x_cstride % 8 != 0) return 0;
  if (N % 8 != 0 || y_cstride % 8 != 0 || y_coff % 8 != 0) return 0;
  if (out_mode != 0 || (res_mode != 0 && res_mode != 1)) return 0;
  if (act != ACT_NONE && act != ACT_RELU && act != ACT_GELU) return 0;
  if ((long long)kh * kw * Cin != K) return 0;
  return 1;
}

extern "C" int dlamd_gemm8(const void* x, const void* w, const float* bias, const void* res, void* y, int B, int H, int W,
                           int Cin, int x_cstride, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int act,
                           int res_mode, int res_cstride, int y_cstride, int y_coff, int y_f32, hipStream_t stream) {
  G8Args a;
  a.x = (const bf16_t*)x;
  a.w = (const bf16_t*)w;
  a.bias = bias;
  a.res = (const bf16_t*)res;
  a.y = y;
  a.H = H; a.W = W; a.Cin = Cin;
  a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw;
  a.Ho = (H + 2 * ph - kh) / sh + 1;
  a.Wo = (W + 2 * pw - kw) / sw + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return -11;
  const long long M = (long long)B * a.Ho * a.Wo;
  a.K = kh * kw * Cin;
  a.N = Cout;
  if (M <= 0 || M > 0x7fffffffLL) return -5;
  a.M = (int)M;
  a.res_mode = res ? res_mode : 0;
  if (!dlamd_gemm8_serves(a.M, a.N, a.K, Cin, x_cstride, kh, kw, act, a.res_mode, 0, y_cstride, y_coff)) return -6;
  if (a.res_mode != 0 && res_cstride % 8 != 0) return -10;
  // 32-bit chunk offsets: every operand must stay below 2^31 16-byte chunks minus slack
  if ((long long)B * H * W * (x_cstride / 8) > 0x3fffffffLL || (long long)Cout * (a.K / 8) > 0x3fffffffLL) return -12;
  a.nkt = a.K / 64;
  a.x_cs8 = x_cstride / 8;
  a.act = act;
  a.res_cstride = res_cstride;
  a.y_cstride = y_cstride;
  a.y_coff = y_coff;
  a.y_f32 = y_f32;
  a.tiles_n = (Cout + 255) / 256;
  const long long tiles = (long long)((a.M + 255) / 256) * a.tiles_n;
  if (tiles > 0x3fffffffLL) return -3;
  a.tiles = (int)tiles;
  const bool is1x1 = kh == 1 && kw == 1 && ph == 0 && pw == 0 && sh == 1 && sw == 1;
  if (!is1x1 && (H >= 32768 || W >= 32768)) return -12;      // the gather packs a row's (hi0, wi0) into 16 bits each
  if (a.res_mode != 0) return is1x1 ? g8_launch<false, true>(a, stream) : g8_launch<true, true>(a, stream);
  return is1x1 ? g8_launch<false, false>(a, stream) : g8_launch<true, false>(a, stream);
}
