// Persistent, software-pipelined bf16 GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   y[M, N] = act(x[M, K] . w[N, K]^T + bias (+ res))          (GEMM: x rows are K-contiguous)
//   y[pixel, Cout] = the same with the row gathered per filter tap (conv: NHWC, Cin % 64 == 0)
//
// This is the kernel behind every plain linear layer of the recognizers (LSTM input projections, ViT QKV /
// proj / fc1 / fc2, decoder projections) and behind the wide 3x3 convolutions of the detector and the CRNN
// (the reference runs those layers inside magic-pdf's CPU torch models, services/ocr_service/ocr_service.py:215-220).
//
// Structure (one 512-thread workgroup per CU, 256 x 256 x 64 tiles, 146 KiB of LDS):
//   * 8 waves = 2 (wr) x 4 (wc).  The tile is cut in two row halves A0/A1 (128 rows each) and two column halves
//     B0/B1.  Wave (wr, wc) owns rows wr*64..+63 of BOTH row halves and columns wc*32..+31 of BOTH column halves,
//     so every wave consumes the four half-tiles in the same order and one K-tile is two intervals:
//         interval 1: (A0 x B0), (A0 x B1)   reads the A0, B0, B1 fragments
//         interval 2: (A1 x B1), (A1 x B0)   reads the A1 fragments
//     32 MFMA 16x16x32 per wave and interval, issued back to back.
//   * Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4), never through a VGPR.  Two half-tiles
//     per K-tile and thread are 8 DMA instructions: A1 of the next K-tile in interval 1 (whose load section also
//     holds 16 of the 24 fragment reads), A0, B0 and B1 of the K-tile two ahead in interval 2.  A0 / B0 / A1 are double buffered, B1 has three
//     slots; every half-tile is in flight for at least two intervals before its first read.  The only wait on the
//     DMA stream is a counted s_waitcnt vmcnt(8) per interval (four half-tiles stay in flight across every barrier).
//   * LDS rows are 128 B (64 bf16); the 16-byte chunk c of row r is stored at chunk c ^ (r & 7), which makes the
//     ds_read_b128 fragment reads conflict-free.  An LDS-DMA writes lane-linear, so the XOR is applied to the
//     per-lane SOURCE address.
//   * The two wave rows run half an interval apart (wr = 1 passes one extra barrier at the start, wr = 0 one at the
//     end): while one group issues its MFMA cluster the other reads fragments, issues DMA and runs epilogues, so
//     the two waves that share a SIMD alternate on its matrix pipe instead of colliding on it.
//   * Persistent over output tiles: the DMA stream simply continues into the next tile, and the epilogue of two
//     quadrants (bias from an LDS copy, residual, activation, v_permlane16_swap so that a lane owns 8 consecutive
//     channels, bf16 pack, one 16-byte store) runs in the NEXT interval's load section, under the other group's
//     MFMA cluster.  Tiles are dealt so that each XCD (private L2) walks consecutive tiles, which share x rows.
//   * MFMA operands are swapped (weights = A operand, x rows = B operand): a lane's four accumulator values are
//     four consecutive output channels of one row.
//
// Synchronisation argument, per half-tile slot (G0 = waves wr 0, G1 = waves wr 1, G1 half an interval behind; every
// load section ends with s_waitcnt vmcnt(8), s_waitcnt lgkmcnt(0), barrier):
//   RAW  a fragment read in interval p needs every wave's DMA into the slot complete: each wave's counted vmcnt in
//        interval p-1 retires it (it is older than the 8 youngest operations there), and both groups' load sections
//        of interval p-1 end at a barrier every wave passes before any wave starts the load section of interval p.
//   WAR  a DMA issued in interval p+1 overwrites a slot last read in interval p: those reads are retired by the
//        lgkmcnt(0) that closes the load section of interval p, ahead of the barrier that G1 must pass before G0
//        can enter the load section of interval p+1 (for G1's own DMA one barrier later).
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
  int x_pad;                    // bytes the x descriptor starts before the tensor (conv: (ph, pw) pixels)
  unsigned x_bytes, w_bytes, y_bytes, r_bytes;    // descriptor extents
  unsigned long long* stamps;   // [grid][8 waves][8] cycle sums, diagnostic builds only
};

namespace {

constexpr int G8_BUF = 49152;                 // double-buffered part of a K-tile: A0 | B0 | A1, 16 KiB each
constexpr int G8_A0 = 0, G8_B0 = 16384, G8_A1 = 32768;
constexpr int G8_B1 = 2 * G8_BUF;             // three B1 slots (K-tile t uses slot t % 3)
constexpr int G8_BIAS = G8_B1 + 3 * 16384;    // two 1 KiB bias images (alternating output tiles)
constexpr int G8_LDS = G8_BIAS + 2048;
constexpr unsigned G8_OOB = 0x80000000u;      // byte offset past any operand (host contract: operands < 2 GiB): reads as zeros

template <int OFF>
__device__ __forceinline__ bf16x8 g8_lds_read(const unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// One LDS-DMA piece: 64 lanes x 16 bytes from base + voff (per lane) + soff (scalar) to lds_wave_base + 16 * lane.
// A buffer load on purpose: the per-lane part of the address is ONE register that stays constant for a whole output
// tile, the K advance is scalar, and a lane whose row does not exist (or whose filter tap falls outside the image)
// carries the out-of-range offset and gets zeros from the descriptor's bounds check - so issuing a piece costs no
// vector instruction.  That matters: the issuing wave shares its SIMD with a wave that is saturating the vector
// issue port with MFMAs, and every VALU instruction in the load section waits for a gap.
__device__ __forceinline__ void g8_dma16(const __amdgpu_buffer_rsrc_t rsrc, const unsigned voff, const int soff, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, soff, 0, 0);
}

// DMA stream position: which K-tile of which output tile the next half-tiles come from, plus this thread's two
// source rows per half-tile the cursor serves (cursor F: A0, B0 and B1; cursor S: A1).
template <bool CONV>
struct G8Cursor {
  int tile, kt, flat;
  int dx, cin0, tap;         // conv: position inside the filter window
  int soff;                  // scalar byte offset of the K-tile in an x row (conv: from the window's first tap)
  unsigned a_off[2];         // byte offset of the two x rows from the descriptor base, G8_OOB when the row does not exist
  unsigned a_mask[2];        // conv: bit t set when filter tap t of the row lies inside the image
  unsigned a_eff[2];         // conv: a_off, or G8_OOB when the current tap is outside
  unsigned b_off[2], b1_off[2];
};

template <bool CONV>
__device__ __forceinline__ void g8_cursor_tap(G8Cursor<CONV>& c) {
  if (CONV) {
#pragma unroll
    for (int j = 0; j < 2; ++j) c.a_eff[j] = ((c.a_mask[j] >> c.tap) & 1u) ? c.a_off[j] : G8_OOB;
  }
}

template <bool CONV, int AH, bool WITH_B>
__device__ __forceinline__ void g8_cursor_rows(G8Cursor<CONV>& c, const G8Args& p, const int rp, const int ch) {
  const bool live = c.tile < p.tiles;
  const int tile = live ? c.tile : 0;
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = tm * 256 + AH * 128 + j * 64 + rp;
    if (WITH_B) {
      const int n = tn * 256 + j * 64 + rp;
      c.b_off[j] = (live && n < p.N) ? (unsigned)(n * (p.K >> 3) + ch) << 4 : G8_OOB;
      c.b1_off[j] = (live && n + 128 < p.N) ? (unsigned)((n + 128) * (p.K >> 3) + ch) << 4 : G8_OOB;
    }
    const bool ok = live && m < p.M;
    if (!CONV) {
      c.a_off[j] = ok ? (unsigned)(m * p.x_cs8 + ch) << 4 : G8_OOB;
    } else {
      const int hw = p.Ho * p.Wo;
      const int mm = ok ? m : 0;
      const int b = mm / hw, rem = mm - b * hw;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int hi0 = ho * p.sh - p.ph, wi0 = wo * p.sw - p.pw;
      // the descriptor base lies (ph, pw) pixels before the tensor, so the window's first tap has a non-negative offset
      c.a_off[j] = ok ? (unsigned)(((b * p.H + ho * p.sh) * p.W + wo * p.sw) * p.x_cs8 + ch) << 4 : G8_OOB;
      const int dx_lo = wi0 < 0 ? -wi0 : 0, dx_hi = p.W - wi0 < p.kw ? p.W - wi0 : p.kw;
      const unsigned xm = dx_hi > dx_lo ? ((1u << dx_hi) - 1u) & ~((1u << dx_lo) - 1u) : 0u;
      unsigned mask = 0;
      for (int dy = 0; dy < p.kh; ++dy)
        if ((unsigned)(hi0 + dy) < (unsigned)p.H) mask |= xm << (dy * p.kw);
      c.a_mask[j] = ok ? mask : 0u;
    }
  }
  g8_cursor_tap<CONV>(c);
}

template <bool CONV, int AH, bool WITH_B>
__device__ __forceinline__ void g8_cursor_advance(G8Cursor<CONV>& c, const G8Args& p, const int stride, const int rp, const int ch) {
  ++c.kt;
  ++c.flat;
  c.soff += 128;
  if (CONV) {
    c.cin0 += 64;
    if (c.cin0 == p.Cin) {                       // next filter tap: the pixel to the right, or the next window row
      c.cin0 = 0;
      ++c.tap;
      c.soff += (p.x_cs8 << 4) - 2 * p.Cin;
      if (++c.dx == p.kw) {
        c.dx = 0;
        c.soff += (p.W - p.kw) * (p.x_cs8 << 4);
      }
      g8_cursor_tap<CONV>(c);
    }
  }
  if (c.kt == p.nkt) {
    c.kt = 0;
    c.dx = c.cin0 = c.tap = c.soff = 0;
    c.tile = c.tile < p.tiles ? c.tile + stride : c.tile;
    g8_cursor_rows<CONV, AH, WITH_B>(c, p, rp, ch);
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
__device__ __forceinline__ void g8_swap16x4(f32x4& a, f32x4& b) {
  asm volatile("s_nop 1\n\t"
               "v_permlane16_swap_b32 %0, %4\n\t"
               "v_permlane16_swap_b32 %1, %5\n\t"
               "v_permlane16_swap_b32 %2, %6\n\t"
               "v_permlane16_swap_b32 %3, %7"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}

// Epilogue of one quadrant (row half AH x column half BH) of the tile at (m0, n0).  The accumulators already hold
// the bias (they start every tile from it), and leave here re-initialised with the bias of the NEXT tile
// (``next_bias``: its LDS image, or nullptr for zeros).
// The two 16-column accumulator tiles of a row are first traded between half-rows of lanes, after which a lane owns
// 8 CONSECUTIVE channels of its row - fq 0 -> 0..7, 1 -> 16..23, 2 -> 8..15, 3 -> 24..31 - so residual and store are
// 16- or 32-byte accesses.  Both go through buffer descriptors: a row past M lies past the descriptor's extent and a
// column past N carries the out-of-range offset, so there is no per-row predicate and no branch.
template <int AH, int BH, bool RES>
__device__ __forceinline__ void g8_epilogue(const G8Args& p, f32x4 (&acc)[2][2][4][2], const int m0, const int n0,
                                            const char* next_bias, const __amdgpu_buffer_rsrc_t srd_y,
                                            const __amdgpu_buffer_rsrc_t srd_r, const int wr, const int wc, const int fr,
                                            const int fq) {
  const int nq = BH * 128 + wc * 32;
  const int n = n0 + nq + (fq & 1) * 16 + (fq >> 1) * 8;                 // first of this lane's 8 columns
  const int mrow = m0 + AH * 128 + wr * 64 + fr;
  const int esz = p.y_f32 ? 4 : 2;
  const unsigned ybase = n < p.N ? (unsigned)(mrow * p.y_cstride + p.y_coff + n) * esz : G8_OOB;
  const unsigned ystep = 16u * p.y_cstride * esz;
  f32x4 nb[2];
  if (next_bias) {
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) nb[nj] = *reinterpret_cast<const f32x4*>(next_bias + (nq + nj * 16 + 4 * fq) * 4);
  } else {
    nb[0] = nb[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32x4 rr[4];
  if (RES) {
    const unsigned rbase = n < p.N ? (unsigned)(mrow * p.res_cstride + n) * 2u : G8_OOB;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      rr[mi] = __builtin_amdgcn_raw_buffer_load_b128(srd_r, (int)(rbase + mi * 32u * p.res_cstride), 0, 0);
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    f32x4 a = acc[AH][BH][mi][0], b = acc[AH][BH][mi][1];
    g8_swap16x4(a, b);
    acc[AH][BH][mi][0] = nb[0];
    acc[AH][BH][mi][1] = nb[1];
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
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
    } else if (p.act == ACT_RELU) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
    }
    const unsigned yoff = ybase + mi * ystep;
    if (p.y_f32) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[0], v[1], v[2], v[3]}), srd_y, (int)yoff, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[4], v[5], v[6], v[7]}), srd_y, (int)yoff + 16, 0, 0);
    } else {
      const bf16x8 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3],
                        (bf16_t)v[4], (bf16_t)v[5], (bf16_t)v[6], (bf16_t)v[7]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), srd_y, (int)yoff, 0, 0);
    }
  }
}

__device__ __forceinline__ unsigned long long g8_clock() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define G8_STAMP(I) if (STAMP) { const unsigned long long t_ = g8_clock(); st[I] += t_ - tprev; tprev = t_; }

#define G8_LOAD_END()                                     \
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_setprio(1);

#define G8_MFMA_END()                                     \
  __builtin_amdgcn_s_setprio(0);                          \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);

#define G8_MFMA_QUAD(AH, BH, WF)                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                            \
  _Pragma("unroll") for (int nj = 0; nj < 2; ++nj)                                                            \
      acc[AH][BH][mi][nj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nj][ks], xf[mi][ks], acc[AH][BH][mi][nj], 0, 0, 0);

// STAMP: diagnostic build only (bench/gemm8_stamp.py): per-wave s_memtime sums of the four sections of each interval
// (load section, wait at the first barrier, MFMA section, wait at the second barrier), written to p.stamps.
template <bool CONV, bool RES, bool STAMP>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const G8Args p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int rp = tid >> 3;                       // DMA role: row rp of each 64-row pass ...
  const int ch = (tid & 7) ^ (rp & 7);           // ... and the logical 16-byte chunk whose swizzled slot is tid & 7

  const int G = gridDim.x;
  const int first_tile = dlamd_xcd_remap(blockIdx.x, G);
  const int my_tiles = first_tile < p.tiles ? (p.tiles - first_tile + G - 1) / G : 0;
  if (my_tiles == 0) return;

  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const unsigned fr_off = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const unsigned xa = lds0 + fr_off + wr * 8192;   // x fragments: row wr*64 + mi*16 + fr of a row half
  const unsigned wa = lds0 + fr_off + wc * 4096;   // w fragments: row wc*32 + nj*16 + fr of a column half
  char* dma_wave = smem + wid * 1024;

  G8Cursor<CONV> cf, cs;                           // F: A0, B0, B1 (two K-tiles ahead), S: A1 (one ahead)
  cf.tile = cs.tile = first_tile;
  cf.kt = cs.kt = cf.flat = cs.flat = 0;
  cf.dx = cf.cin0 = cf.tap = cf.soff = cs.dx = cs.cin0 = cs.tap = cs.soff = 0;
  const __amdgpu_buffer_rsrc_t srd_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.x)) - p.x_pad, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t srd_r = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16_t*>(p.res ? p.res : p.x), 0, p.res ? p.r_bytes : 0, 0x00020000);
  g8_cursor_rows<CONV, 0, true>(cf, p, rp, ch);
  g8_cursor_rows<CONV, 1, false>(cs, p, rp, ch);

  auto bias_dma = [&](const int tile_, const int which) {
    if (p.bias != nullptr && wid < 4) {
      const int tn = (tile_ < p.tiles ? tile_ : first_tile) % p.tiles_n;
      int n = tn * 256 + wid * 64 + lane;
      n = n < p.N ? n : p.N - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + n),
                                       (__attribute__((address_space(3))) void*)(smem + G8_BIAS + which * 1024 + wid * 256), 4, 0, 0);
    }
  };
#define G8_STAGE_A(C, SLOT)                                                                             \
  {                                                                                                     \
    char* d = dma_wave + ((C.flat & 1) ? G8_BUF : 0) + SLOT;                                            \
    g8_dma16(srd_x, CONV ? C.a_eff[0] : C.a_off[0], C.soff, d);                                         \
    g8_dma16(srd_x, CONV ? C.a_eff[1] : C.a_off[1], C.soff, d + 8192);                                  \
  }
#define G8_STAGE_B0(C)                                                                                  \
  {                                                                                                     \
    char* d = dma_wave + ((C.flat & 1) ? G8_BUF : 0) + G8_B0;                                           \
    g8_dma16(srd_w, C.b_off[0], C.kt * 128, d);                                                         \
    g8_dma16(srd_w, C.b_off[1], C.kt * 128, d + 8192);                                                  \
  }
#define G8_STAGE_B1(C)                                                                                  \
  {                                                                                                     \
    char* d = dma_wave + G8_B1 + (C.flat % 3) * 16384;                                                  \
    g8_dma16(srd_w, C.b1_off[0], C.kt * 128, d);                                                        \
    g8_dma16(srd_w, C.b1_off[1], C.kt * 128, d + 8192);                                                 \
  }

  // ---- prologue: bias images of the first two tiles; K-tile 0 completely, A0 + B0 + B1 of K-tile 1
  bias_dma(first_tile, 0);
  bias_dma(first_tile + G, 1);
  G8_STAGE_A(cf, G8_A0);
  G8_STAGE_B0(cf);
  G8_STAGE_B1(cf);
  G8_STAGE_A(cs, G8_A1);
  g8_cursor_advance<CONV, 1, false>(cs, p, G, rp, ch);
  g8_cursor_advance<CONV, 0, true>(cf, p, G, rp, ch);
  G8_STAGE_A(cf, G8_A0);
  G8_STAGE_B0(cf);
  G8_STAGE_B1(cf);
  g8_cursor_advance<CONV, 0, true>(cf, p, G, rp, ch);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // bias images, A0, B0 and B1 of K-tile 0 have landed
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // accumulators start from the bias (pre-swap layout: channel BH*128 + wc*32 + nj*16 + 4 fq + q)
  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *reinterpret_cast<const f32x4*>(smem + G8_BIAS + (b * 128 + wc * 32 + j * 16 + 4 * fq) * 4);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[a][b][i][j] = bv;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 1) __builtin_amdgcn_s_barrier();                  // G1 runs half an interval behind G0
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 xf[4][2], w0[2][2], w1[2][2];
  int tile = first_tile, kt = 0, tile_seq = 0;
  int m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * 256;
  int pm0 = 0, pn0 = 0;
  const int total = my_tiles * p.nkt;
  int slot3 = 0;                                              // T % 3

  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
  if (STAMP) tprev = g8_clock();
  for (int T = 0; T < total; ++T) {
    const unsigned boff = (T & 1) ? G8_BUF : 0;
    const bool first = kt == 0, last = kt == p.nkt - 1;
    const unsigned xb = xa + boff, wb = wa + boff;
    const unsigned xb1 = xb ^ 64, wb1 = wb ^ 64;
    const unsigned w1b = wa + G8_B1 + slot3 * 16384, w1b1 = w1b ^ 64;
    // bias image of tile s is slot s & 1; quadrants finished in tile s re-start from the bias of tile s + 1
    const char* cur_bias = p.bias ? smem + G8_BIAS + (tile_seq & 1) * 1024 : nullptr;
    const char* nxt_bias = p.bias ? smem + G8_BIAS + ((tile_seq ^ 1) & 1) * 1024 : nullptr;

    // ---------------- interval 1: A0 x B0, A0 x B1
    if (first && T > 0) {
      g8_epilogue<1, 1, RES>(p, acc, pm0, pn0, cur_bias, srd_y, srd_r, wr, wc, fr, fq);
      g8_epilogue<1, 0, RES>(p, acc, pm0, pn0, cur_bias, srd_y, srd_r, wr, wc, fr, fq);
      bias_dma(tile + G, (tile_seq ^ 1) & 1);   // the image of the previous tile is free: its last readers ran a tile ago
    }
    xf[0][0] = g8_lds_read<G8_A0 + 0 * 2048>(xb); xf[0][1] = g8_lds_read<G8_A0 + 0 * 2048>(xb1);
    xf[1][0] = g8_lds_read<G8_A0 + 1 * 2048>(xb); xf[1][1] = g8_lds_read<G8_A0 + 1 * 2048>(xb1);
    xf[2][0] = g8_lds_read<G8_A0 + 2 * 2048>(xb); xf[2][1] = g8_lds_read<G8_A0 + 2 * 2048>(xb1);
    xf[3][0] = g8_lds_read<G8_A0 + 3 * 2048>(xb); xf[3][1] = g8_lds_read<G8_A0 + 3 * 2048>(xb1);
    w0[0][0] = g8_lds_read<G8_B0>(wb); w0[0][1] = g8_lds_read<G8_B0>(wb1);
    w0[1][0] = g8_lds_read<G8_B0 + 2048>(wb); w0[1][1] = g8_lds_read<G8_B0 + 2048>(wb1);
    w1[0][0] = g8_lds_read<0>(w1b); w1[0][1] = g8_lds_read<0>(w1b1);
    w1[1][0] = g8_lds_read<2048>(w1b); w1[1][1] = g8_lds_read<2048>(w1b1);
    G8_STAMP(0);
    G8_STAGE_A(cs, G8_A1);                                    // A1 of K-tile T+1
    g8_cursor_advance<CONV, 1, false>(cs, p, G, rp, ch);
    G8_STAMP(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    G8_STAMP(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G8_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    G8_STAMP(4);
    __builtin_amdgcn_s_setprio(1);
    G8_MFMA_QUAD(0, 0, w0);
    G8_MFMA_QUAD(0, 1, w1);
    __builtin_amdgcn_s_setprio(0);
    G8_STAMP(5);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    G8_STAMP(6);

    // ---------------- interval 2: A1 x B1, A1 x B0
    if (last) {
      g8_epilogue<0, 0, RES>(p, acc, m0, n0, nxt_bias, srd_y, srd_r, wr, wc, fr, fq);
      g8_epilogue<0, 1, RES>(p, acc, m0, n0, nxt_bias, srd_y, srd_r, wr, wc, fr, fq);
    }
    xf[0][0] = g8_lds_read<G8_A1 + 0 * 2048>(xb); xf[0][1] = g8_lds_read<G8_A1 + 0 * 2048>(xb1);
    xf[1][0] = g8_lds_read<G8_A1 + 1 * 2048>(xb); xf[1][1] = g8_lds_read<G8_A1 + 1 * 2048>(xb1);
    xf[2][0] = g8_lds_read<G8_A1 + 2 * 2048>(xb); xf[2][1] = g8_lds_read<G8_A1 + 2 * 2048>(xb1);
    xf[3][0] = g8_lds_read<G8_A1 + 3 * 2048>(xb); xf[3][1] = g8_lds_read<G8_A1 + 3 * 2048>(xb1);
    G8_STAGE_A(cf, G8_A0);                                    // A0, B0 and B1 of K-tile T+2
    G8_STAGE_B0(cf);
    G8_STAGE_B1(cf);
    g8_cursor_advance<CONV, 0, true>(cf, p, G, rp, ch);
    G8_LOAD_END();
    G8_MFMA_QUAD(1, 1, w1);
    G8_MFMA_QUAD(1, 0, w0);
    G8_MFMA_END();
    G8_STAMP(7);

    slot3 = slot3 == 2 ? 0 : slot3 + 1;
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
  g8_epilogue<1, 1, RES>(p, acc, pm0, pn0, nullptr, srd_y, srd_r, wr, wc, fr, fq);
  g8_epilogue<1, 0, RES>(p, acc, pm0, pn0, nullptr, srd_y, srd_r, wr, wc, fr, fq);
  if (STAMP && lane == 0 && p.stamps) {
#pragma unroll
    for (int i = 0; i < 8; ++i) p.stamps[((long long)blockIdx.x * 8 + wid) * 8 + i] = st[i];
  }
  __builtin_amdgcn_sched_barrier(0);
  if (wr == 0) __builtin_amdgcn_s_barrier();                  // balances G1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no LDS-DMA may outlive the workgroup's LDS allocation
}

#undef G8_STAGE_A
#undef G8_STAGE_B0
#undef G8_STAGE_B1

template <bool CONV, bool RES, bool STAMP = false>
int g8_launch(const G8Args& a, hipStream_t stream) {
  auto kern = gemm8p_kernel<CONV, RES, STAMP>;
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
  // the DMA addresses rows by 32-bit byte offsets from a buffer descriptor; 2^31 marks a row that does not exist
  a.x_pad = (ph * W + pw) * x_cstride * 2;
  const long long xb = ((long long)B * H * W - 1) * x_cstride * 2 + (long long)Cin * 2 + a.x_pad;
  const long long wb = (long long)Cout * a.K * 2;
  if (xb >= 0x7fffffffLL || wb >= 0x7fffffffLL || kh * kw > 32) return -12;
  const long long yb = (long long)a.M * y_cstride * (y_f32 ? 4 : 2);
  const long long rb = a.res_mode ? (long long)a.M * res_cstride * 2 : 0;
  if (yb >= 0x7fffffffLL || rb >= 0x7fffffffLL) return -12;
  a.x_bytes = (unsigned)xb;
  a.w_bytes = (unsigned)wb;
  a.y_bytes = (unsigned)yb;
  a.r_bytes = (unsigned)rb;
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
  a.stamps = nullptr;
  const bool is1x1 = kh == 1 && kw == 1 && ph == 0 && pw == 0 && sh == 1 && sw == 1;
  if (!is1x1 && (H >= 32768 || W >= 32768)) return -12;      // the gather packs a row's (hi0, wi0) into 16 bits each
  if (a.res_mode != 0) return is1x1 ? g8_launch<false, true>(a, stream) : g8_launch<true, true>(a, stream);
  return is1x1 ? g8_launch<false, false>(a, stream) : g8_launch<true, false>(a, stream);
}

// Diagnostic: plain GEMM (no bias) through the stamped instantiation; stamps = [min(tiles, CUs)][8][8] u64.
extern "C" int dlamd_gemm8_stamp(const void* x, const void* w, void* y, int M, int N, int K, void* stamps, hipStream_t stream) {
  G8Args a = {};
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = y;
  a.M = M; a.N = N; a.K = K; a.nkt = K / 64; a.x_cs8 = K / 8;
  a.H = 1; a.W = M; a.Cin = K; a.Ho = 1; a.Wo = M; a.kh = a.kw = a.sh = a.sw = 1;
  a.y_cstride = N;
  a.tiles_n = (N + 255) / 256;
  a.tiles = ((M + 255) / 256) * a.tiles_n;
  a.stamps = (unsigned long long*)stamps;
  a.x_pad = 0; a.x_bytes = (unsigned)((long long)M * K * 2); a.w_bytes = (unsigned)((long long)N * K * 2);
  a.y_bytes = (unsigned)((long long)M * N * 2); a.r_bytes = 0;
  if (!dlamd_gemm8_serves(M, N, K, K, K, 1, 1, 0, 0, 0, N, 0)) return -6;
  return g8_launch<false, false, true>(a, stream);
}
