// Implicit-GEMM convolution / GEMM on MFMA for gfx950 (MI355X).
//
// One kernel family serves every GEMM-shaped op of the OCR stack:
//   * DBNet backbone / FPN / head convolutions (NHWC bf16, any kh x kw, stride, pad)
//   * 1x1 convolutions and plain GEMMs  y[M,N] = x[M,K] . w[N,K]^T  (H=1, W=M)
//   * transposed conv 2x2 stride 2 (GEMM + pixel-shuffle store)
//   * FPN "nearest-upsample + add" fused on the residual read, and
//     "upsample xf + concat" fused on the store
//   * bias + ReLU / hardswish / GELU / sigmoid epilogues; the DB head's
//     sigmoid + threshold epilogue emits probability map and bitmask.
//
// Design (CDNA4): 256 threads = 4 wave64; block tile BM pixels x BN channels,
// K-step 64 (128-byte LDS rows, XOR-swizzled in 16-byte chunks so that the
// ds_read_b128 fragment reads spread over all banks).  Operands are swapped
// into the MFMA (weights as the A operand, pixels as the B operand) so that a
// lane's four accumulator values are four CONSECUTIVE output channels of one
// pixel -> 8-byte packed bf16 stores in NHWC without an LDS transpose.
// Global->LDS staging is register-staged with issue-early / write-late
// (loads of K-tile t+1 are in flight while tile t is multiplied), one barrier
// per K-tile, LDS double buffered.  Tile ids are remapped so that each XCD
// (private L2) walks a contiguous range of output tiles.
#include <stdlib.h>

#include "common.h"

struct ConvArgs {
  const bf16_t* x;
  const bf16_t* w;      // [Cout][kh][kw][Cin]  (K contiguous)
  const float* bias;    // [Cout] or nullptr
  const bf16_t* res;    // residual or nullptr
  void* y;              // bf16 (or f32 when y_f32)
  uint8_t* mask;        // only out_mode 3
  int B, H, W, Cin, x_cstride;
  int Ho, Wo, Cout;
  int kh, kw, sh, sw, ph, pw;
  int K, M;
  int act;
  int res_mode;     // 0 none, 1 same resolution, 2 nearest-2x upsampled source
  int res_cstride;
  int out_mode;     // 0 plain, 1 pixel-shuffle 2x2, 2 replicate fxf, 3 prob-map (Cout==4 -> 2x2 pixels)
  int out_f;
  int y_cstride, y_coff;
  int y_f32;
  float thr;
  int tiles_m, tiles_n;
  int nbuf;         // LDS operand buffers: 2 (double buffered K loop) or 1 when the whole K fits one tile
  const bf16_t* wfrag;  // weights pre-swizzled to MFMA fragment order [N/16][Kpad/32][64][8] (or nullptr)
  int k32;          // Kpad / 32
  int n16;          // padded Cout / 16: 16-row weight tiles that exist in wfrag (tile reads are clamped to it)
  int coalesce;     // plain bf16 store whose rows are 16-byte aligned: stage the output tile in LDS, store full lines
};

static constexpr int BK = 64;  // K elements per tile (128 B per LDS row)

// Fused epilogue for one output pixel m held by this lane: bias + residual (+ nearest-2x upsampled residual) +
// activation, then one of the store modes (plain / concat offset, pixel-shuffle, replicate, prob-map + mask).
// acc[i][j][q] is channel nbase + i*16 + fq*4 + q of that pixel.
// Where the epilogue puts a plain bf16 result: straight to global memory (8 bytes per lane: a wave instruction is
// 16 pixel rows x 32 bytes, i.e. 16 partial-line requests) or into an LDS image of the output tile that the
// workgroup then writes out as whole 16-byte-per-lane rows.  The partial-line form costs ~1.25 ps per output
// element whatever the K depth - 25-30 % of an 18-K-tile layer (bench/conv_ktile_probe.py).
struct ConvLdsSink {
  char* base;      // LDS image [rows][rowb bytes], nullptr: store to global
  int m0, n0, rowb;
};

template <int FN, int FM>
__device__ __forceinline__ void conv_epilogue_pixel(const ConvArgs& p, const f32x4 (&acc)[FN][FM], const int j,
                                                    const int m, const int nbase, const int fq,
                                                    const ConvLdsSink sink = ConvLdsSink{nullptr, 0, 0, 0}) {
  // Plain bf16 store with bias, optional same-size residual and none / ReLU (most layers): decided once per pixel
  // instead of per value - the general path below costs several hundred VALU instructions per tile and wave.
  if (p.out_mode == 0 && !p.y_f32 && p.res_mode <= 1 && (p.act == ACT_NONE || p.act == ACT_RELU) &&
      (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) {
    const bool relu = p.act == ACT_RELU;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = nbase + i * 16 + fq * 4;
      if (n >= p.Cout) continue;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
      float v0 = acc[i][j][0] + bv[0], v1 = acc[i][j][1] + bv[1], v2 = acc[i][j][2] + bv[2], v3 = acc[i][j][3] + bv[3];
      if (p.res_mode == 1) {
        const u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + (long long)m * p.res_cstride + n);
        v0 += bf16_bits_to_f32(rr[0] & 0xffffu);
        v1 += bf16_bits_to_f32(rr[0] >> 16);
        v2 += bf16_bits_to_f32(rr[1] & 0xffffu);
        v3 += bf16_bits_to_f32(rr[1] >> 16);
      }
      if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
      const bf16x4 o = {(bf16_t)v0, (bf16_t)v1, (bf16_t)v2, (bf16_t)v3};
      if (sink.base) *reinterpret_cast<bf16x4*>(sink.base + (m - sink.m0) * sink.rowb + (n - sink.n0) * 2) = o;
      else *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(p.y) + (long long)m * p.y_cstride + p.y_coff + n) = o;
    }
    return;
  }
  const int hw = p.Ho * p.Wo;
  int b = 0, ho = 0, wo = 0;
  if (p.out_mode != 0 || p.res_mode == 2) {
    b = m / hw;
    int rem = m - b * hw;
    ho = rem / p.Wo;
    wo = rem - ho * p.Wo;
  }
  long long res_pix = m;
  if (p.res_mode == 2) {
    int Hr = (p.Ho + 1) >> 1, Wr = (p.Wo + 1) >> 1;
    res_pix = ((long long)b * Hr + (ho >> 1)) * Wr + (wo >> 1);
  }
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int n = nbase + i * 16 + fq * 4;
    if (n >= p.Cout) continue;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float t = acc[i][j][q];
      if (p.bias) t += p.bias[n + q];
      v[q] = t;
    }
    if (p.res_mode != 0) {
      u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + res_pix * p.res_cstride + n);
      v[0] += bf16_bits_to_f32(rr[0] & 0xffffu);
      v[1] += bf16_bits_to_f32(rr[0] >> 16);
      v[2] += bf16_bits_to_f32(rr[1] & 0xffffu);
      v[3] += bf16_bits_to_f32(rr[1] >> 16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = dlamd_act(v[q], p.act);

    if (p.out_mode == 3) {
      // Cout == 4: the four values are the 2x2 output pixels of a 2x2/s2 deconv.
      if (n == 0) {
        bf16_t* yp = reinterpret_cast<bf16_t*>(p.y);
        const int H2 = p.Ho * 2, W2 = p.Wo * 2;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          long long pix = ((long long)b * H2 + 2 * ho + dy) * W2 + 2 * wo;
          bf16x2 pr = {(bf16_t)v[dy * 2], (bf16_t)v[dy * 2 + 1]};
          *reinterpret_cast<bf16x2*>(yp + pix) = pr;
          // threshold the ROUNDED value so mask and stored probability always agree
          uint16_t mk = uint16_t((float)pr[0] > p.thr ? 1 : 0) | uint16_t(((float)pr[1] > p.thr ? 1 : 0) << 8);
          *reinterpret_cast<uint16_t*>(p.mask + pix) = mk;
        }
      }
      continue;
    }

    if (p.y_f32) {
      float* yp = reinterpret_cast<float*>(p.y);
      *reinterpret_cast<f32x4*>(yp + (long long)m * p.y_cstride + p.y_coff + n) = f32x4{v[0], v[1], v[2], v[3]};
      continue;
    }
    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    bf16_t* yp = reinterpret_cast<bf16_t*>(p.y);
    if (p.out_mode == 0) {
      if (sink.base) *reinterpret_cast<bf16x4*>(sink.base + (m - sink.m0) * sink.rowb + (n - sink.n0) * 2) = o;
      else *reinterpret_cast<bf16x4*>(yp + (long long)m * p.y_cstride + p.y_coff + n) = o;
    } else if (p.out_mode == 1) {
      const int Cp = p.Cout >> 2;
      const int g = n / Cp, cc = n - g * Cp;
      const int dy = g >> 1, dx = g & 1;
      long long pix = ((long long)b * (2 * p.Ho) + 2 * ho + dy) * (2 * p.Wo) + 2 * wo + dx;
      *reinterpret_cast<bf16x4*>(yp + pix * p.y_cstride + p.y_coff + cc) = o;
    } else {
      const int f = p.out_f;
      for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
          long long pix = ((long long)b * (f * p.Ho) + f * ho + dy) * (f * p.Wo) + f * wo + dx;
          *reinterpret_cast<bf16x4*>(yp + pix * p.y_cstride + p.y_coff + n) = o;
        }
    }
  }
}

// Writes the LDS image of a BM x BN bf16 output tile to global memory: 16 bytes per lane, consecutive lanes along
// the channel row, so a wave instruction covers whole 128-byte lines.  All 256 threads call it after a barrier.
template <int BM, int BN>
__device__ __forceinline__ void conv_store_tile_from_lds(const ConvArgs& p, const char* img, const int rowb, const int m0,
                                                         const int n0, const int tid) {
  constexpr int CPR = BN / 8;                 // 16-byte chunks per row
  bf16_t* yp = reinterpret_cast<bf16_t*>(p.y);
#pragma unroll
  for (int q = tid; q < BM * CPR; q += 256) {
    const int row = q / CPR, cc = q - row * CPR;
    const int m = m0 + row, n = n0 + cc * 8;
    if (m >= p.M || n >= p.Cout) continue;
    const u32x4 v = *reinterpret_cast<const u32x4*>(img + row * rowb + cc * 16);
    bf16_t* dst = yp + (long long)m * p.y_cstride + p.y_coff + n;
    if (n + 8 <= p.Cout) *reinterpret_cast<u32x4*>(dst) = v;
    else *reinterpret_cast<u32x2*>(dst) = u32x2{v[0], v[1]};      // Cout % 8 == 4: the last chunk is half valid
  }
}

template <int BM, int BN, int WM, int WN, bool IS_1X1, bool B_DIRECT, int BKT, bool PERSIST>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& p) {
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  // K tile of BKT elements: LDS rows of ROWB bytes = CH 16-byte chunks, XOR-swizzled with the row index;
  // 256 threads cover RP rows per pass.  BKT = 128 halves the barriers (and exposed load latencies) per FLOP.
  constexpr int ROWB = BKT * 2, CH = BKT / 8, RP = 256 / CH, KS = BKT / 32;
  constexpr int PA = (BM + RP - 1) / RP, PB = (BN + RP - 1) / RP;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TM % 16 == 0 && TN % 16 == 0, "tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // B_DIRECT: the weight operand never touches LDS.  Weights are small and L2-resident; the host pre-swizzles
  // them into MFMA fragment order so a wave reads its A-operand fragments as contiguous 1 KiB loads, one
  // K-tile ahead.  That halves the LDS footprint of a block (twice the blocks per CU - this kernel lives on
  // occupancy to hide global-load latency) and removes the weight LDS write + read passes.
  constexpr int LB = B_DIRECT ? 0 : BN;
  char* As = smem;                            // nbuf x BM x ROWB
  char* Bs = smem + p.nbuf * BM * ROWB;        // nbuf x BN x 128 B (unused with B_DIRECT)
  // the tap table sits behind the operand buffers; the persistent form needs it again after the epilogue has
  // used the front of LDS as the output image, so there it sits behind whichever of the two is larger
  constexpr int OIMG = BM * (BN * 2 + 16);
  const int opnd = p.nbuf * (BM + LB) * ROWB;
  uint32_t* ktab = reinterpret_cast<uint32_t*>(smem + ((PERSIST && opnd < OIMG) ? OIMG : opnd));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;

  const int nwg = gridDim.x;
  const int nkt = (p.K + BKT - 1) / BKT;

  const int c = tid % CH;       // 16-byte chunk inside the K row
  const int r0 = tid / CH;      // 0..RP-1

  if (!IS_1X1) {
    // (tap, cin) decomposition of every 8-wide K chunk, once per block.
    for (int e = tid; e < nkt * CH; e += 256) {
      int k = e * 8;
      uint32_t t = 0xFFFFFFFFu, off = 0;
      if (k < p.K) {
        int tap = k / p.Cin;
        int cin0 = k - tap * p.Cin;
        int dy = tap / p.kw, dx = tap - dy * p.kw;
        t = (uint32_t(dy) << 8) | uint32_t(dx);
        off = uint32_t((dy * p.W + dx) * p.x_cstride + cin0);
      }
      ktab[2 * e] = t;
      ktab[2 * e + 1] = off;
    }
    __syncthreads();
  }

  // PERSIST: the workgroup walks tiles t, t + G, ... (grid = resident workgroups): the tap table above is built
  // once, and a finished tile's stores drain while the next tile starts instead of holding a dead wave slot.
  const int ntiles = p.tiles_m * p.tiles_n;
  for (int tile = dlamd_xcd_remap(blockIdx.x, nwg); tile < ntiles; tile += nwg) {
  // (PERSIST) thread-index-derived values are re-derived per tile from a laundered id, so that none of them has
  // to stay live across the previous tile's epilogue - that is what kept the kernel above 128 registers
  int tid_l = threadIdx.x;
  if (PERSIST) asm volatile("" : "+v"(tid_l));
  const int tid = tid_l, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int c = tid % CH, r0 = tid / CH;
  const int tn = tile % p.tiles_n;
  const int tm = tile / p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  // Per-thread row descriptors (fixed across the K loop).
  long long a_base[PA];
  int a_hi0[PA], a_wi0[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int r = r0 + RP * i;
    int m = m0 + r;
    bool ok = (r < BM) && (m < p.M);
    int mm = ok ? m : 0;
    int hw = p.Ho * p.Wo;
    int b = mm / hw;
    int rem = mm - b * hw;
    int ho = rem / p.Wo;
    int wo = rem - ho * p.Wo;
    int hi0 = ho * p.sh - p.ph, wi0 = wo * p.sw - p.pw;
    a_ok[i] = ok;
    a_hi0[i] = hi0;
    a_wi0[i] = wi0;
    a_base[i] = ((long long)(b * p.H + hi0) * p.W + wi0) * p.x_cstride;
  }
  long long b_base[PB];
  bool b_ok[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    int r = r0 + RP * i;
    int n = n0 + r;
    b_ok[i] = (r < BN) && (n < p.Cout);
    b_base[i] = (long long)(b_ok[i] ? n : 0) * p.K;
  }

  u32x4 a_reg[PA], b_reg[PB];

  auto load_tile = [&](int kt) {
    int k = kt * BKT + c * 8;
    bool kok;
    int dy = 0, dx = 0;
    long long off;
    if (IS_1X1) {
      kok = k < p.K;
      off = k;
    } else {
      uint32_t t = ktab[2 * (kt * CH + c)];
      kok = t != 0xFFFFFFFFu;
      dy = int(t >> 8);
      dx = int(t & 0xff);
      off = (long long)ktab[2 * (kt * CH + c) + 1];
    }
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      bool v = a_ok[i] && kok;
      if (!IS_1X1) {
        int hi = a_hi0[i] + dy, wi = a_wi0[i] + dx;
        v = v && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      }
      u32x4 val = {0u, 0u, 0u, 0u};
      if (v) val = *reinterpret_cast<const u32x4*>(p.x + a_base[i] + off);
      a_reg[i] = val;
    }
    if (!B_DIRECT) {
      bool kwok = k < p.K;
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        u32x4 val = {0u, 0u, 0u, 0u};
        if (b_ok[i] && kwok) val = *reinterpret_cast<const u32x4*>(p.w + b_base[i] + k);
        b_reg[i] = val;
      }
    }
  };

  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int r = r0 + RP * i;
      if (BM % RP == 0 || r < BM)
        *reinterpret_cast<u32x4*>(As + buf * BM * ROWB + r * ROWB + ((c ^ (r & (CH - 1))) << 4)) = a_reg[i];
    }
    if (!B_DIRECT) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        int r = r0 + RP * i;
        if (BN % RP == 0 || r < BN)
          *reinterpret_cast<u32x4*>(Bs + buf * BN * ROWB + r * ROWB + ((c ^ (r & (CH - 1))) << 4)) = b_reg[i];
      }
    }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fq = lane >> 4;

  // direct weight fragments: wcur = tile kt, wnxt = tile kt+1 (in flight during the MFMAs of tile kt)
  bf16x8 wcur[KS][FN], wnxt[KS][FN];
  const bf16_t* wfp = nullptr;
  long long wtile[FN];
  if (B_DIRECT) {
    wfp = p.wfrag + (long long)lane * 8;
    // a tile wider than the padded weight matrix (BN = 256 on a 128-padded Cout) re-reads the last tile
    // instead of running off the buffer; those channels are >= Cout and never stored
#pragma unroll
    for (int i = 0; i < FN; ++i) wtile[i] = (long long)min(n0 / 16 + wn * FN + i, p.n16 - 1) * p.k32 * 512;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < FN; ++i)
        wcur[ks][i] = *reinterpret_cast<const bf16x8*>(wfp + wtile[i] + (long long)ks * 512);
  }

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      load_tile(kt + 1);
      if (B_DIRECT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < FN; ++i)
            wnxt[ks][i] = *reinterpret_cast<const bf16x8*>(wfp + wtile[i] + (long long)((kt + 1) * KS + ks) * 512);
      }
    }
    const char* Ab = As + buf * BM * ROWB;
    const char* Bb = Bs + buf * BN * ROWB;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 wf[FN], xf[FM];
      const int chunk = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        if (B_DIRECT) {
          wf[i] = wcur[ks][i];
        } else {
          int row = wn * TN + i * 16 + frow;
          wf[i] = *reinterpret_cast<const bf16x8*>(Bb + row * ROWB + ((chunk ^ (row & (CH - 1))) << 4));
        }
      }
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        int row = wm * TM + j * 16 + frow;
        xf[j] = *reinterpret_cast<const bf16x8*>(Ab + row * ROWB + ((chunk ^ (row & (CH - 1))) << 4));
      }
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      store_tile(buf ^ 1);
      if (B_DIRECT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < FN; ++i) wcur[ks][i] = wnxt[ks][i];
      }
    }
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // acc[i][j][q]: channel n0 + wn*TN + i*16 + fq*4 + q, pixel m0 + wm*TM + j*16 + frow
  if constexpr (FM <= 4) {
    // the K loop ended on a barrier: the operand buffers are free and become the output image
    constexpr int OROWB = BN * 2 + 16;          // padded rows: the 16 pixel rows of a fragment spread over the banks
    const ConvLdsSink sink{p.coalesce ? smem : nullptr, m0, n0, OROWB};
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0 + wm * TM + j * 16 + frow;
      if (m >= p.M) continue;
      conv_epilogue_pixel<FN, FM>(p, acc, j, m, n0 + wn * TN, fq, sink);
    }
    if (p.coalesce) {
      __syncthreads();
      conv_store_tile_from_lds<BM, BN>(p, smem, OROWB, m0, n0, tid);
    }
  } else {
    // tall per-wave tiles: rolled loop + accumulator rotation (an unrolled epilogue of this size pushes acc[][j]
    // into scratch, and a kernel with scratch runs 4-5x slower)
#pragma nounroll
    for (int j = 0; j < FM; ++j) {
      const int m = m0 + wm * TM + j * 16 + frow;
      if (m < p.M) conv_epilogue_pixel<FN, FM>(p, acc, 0, m, n0 + wn * TN, fq);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int jj = 0; jj + 1 < FM; ++jj) acc[i][jj] = acc[i][jj + 1];
    }
  }
  if constexpr (!PERSIST) break;    // one tile per workgroup: no loop for the compiler to reason about
  __syncthreads();                  // the output image / operand buffers are reused by the next tile
  }
}

template <int BM, int BN, int WM, int WN, bool IS_1X1, bool B_DIRECT, int BKT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
  conv_igemm_body<BM, BN, WM, WN, IS_1X1, B_DIRECT, BKT, false>(p);
}

// persistent form: pinned to 4 waves per SIMD (<= 128 registers) - the tile loop otherwise tempts the compiler
// into keeping loop invariants live (147-153 registers, 3 waves), and this kernel lives on resident workgroups
template <int BM, int BN, int WM, int WN, bool IS_1X1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_igemm_persistent_kernel(const ConvArgs p) {
  conv_igemm_body<BM, BN, WM, WN, IS_1X1, true, 64, true>(p);
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 convolution with an LDS-resident input PATCH (stride 1 along x, stride 1 or 2 along y, pad 1).
//
// The generic kernel above re-gathers every input pixel once per filter tap (a 9x im2col expansion of the
// L2 -> LDS traffic) and pays one barrier + one global-load latency per 64-deep K tile.  For the 3x3 layers
// that carry most of the FLOPs (DBNet stages / FPN smooth convs / head, CRNN c3-c5) this kernel instead
// stages the (TH*SH + 2) x (TW + 2) input patch of a TH x TW output tile ONCE per 64-channel chunk and forms
// all nine taps from LDS by offsetting the fragment reads:
//   * L2 -> LDS traffic drops from 9x to ~1.4x of the input,
//   * one barrier pair per 64 input channels = per 18 MFMA k-steps instead of per 2,
//   * weights never touch LDS: they arrive in MFMA fragment order straight from L2 (see B_DIRECT above),
//     prefetched one k-step ahead.
// Tile shape is chosen per layer so feature maps that are only a few rows high (CRNN: 12, 6, 3 rows) still
// fill the 128-pixel tile: 8x16, 4x32, 2x64 or 1x128.
// Pixel (py, px) of the tile is fragment row lp = py*TW + px; with TW a multiple of 16 a wave's 16 fragment
// rows are 16 consecutive x positions, so tap (dy, dx) is a constant offset in the patch.
template <int TH, int TW, int SH, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void conv3x3_patch_kernel(const ConvArgs p) {
  constexpr int BM = TH * TW;
  constexpr int NT = WM * WN * 64;                  // 4 waves on a 128-pixel tile, 8 waves on a 256-pixel tile
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int PH = (TH - 1) * SH + 3, PW = TW + 2;
  constexpr int NCH = PH * PW * 8;                  // 16-byte chunks of one 64-channel patch
  constexpr int PER = (NCH + NT - 1) / NT;
  static_assert((WM * WN == 4 && BM == 128) || (WM * WN == 8 && BM == 256), "tile");
  static_assert(TW % 16 == 0, "tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // PH*PW pixels x 128 B, XOR-swizzled chunks

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int frow = lane & 15, fq = lane >> 4;

  // tile decomposition: blockIdx.x = n-tile fastest (blocks sharing a patch run back to back), then x, y, image
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  int t = dlamd_xcd_remap(blockIdx.x, gridDim.x);
  const int tn = t % p.tiles_n; t /= p.tiles_n;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int n0 = tn * BN;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * SH - 1, ix0 = ox0 - 1;
  const int nch = p.Cin >> 6;                        // 64-channel chunks (host guarantees Cin % 64 == 0)

  // per-thread patch chunks (fixed across the channel loop)
  long long g_off[PER];
  int l_off[PER];
  bool g_ok[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + i * NT;
    const int pp = e >> 3, c = e & 7;
    const int pr = pp / PW, pc = pp - pr * PW;
    const int y = iy0 + pr, x = ix0 + pc;
    g_ok[i] = e < NCH && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    g_off[i] = ((long long)(b * p.H + (g_ok[i] ? y : 0)) * p.W + (g_ok[i] ? x : 0)) * p.x_cstride + c * 8;
    l_off[i] = e < NCH ? pp * 128 + ((c ^ (pp & 7)) << 4) : -1;
  }
  u32x4 preg[PER];
  auto load_patch = [&](int cc) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (g_ok[i]) v = *reinterpret_cast<const u32x4*>(p.x + g_off[i] + cc * 64);
      preg[i] = v;
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (l_off[i] >= 0) *reinterpret_cast<u32x4*>(smem + l_off[i]) = preg[i];
  };

  // fragment-row -> patch pixel (tap 0,0) for each of this wave's FM row tiles
  int pbase[FM];
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int lp = wm * TM + j * 16 + frow;
    const int py = lp / TW, px = lp - py * TW;
    pbase[j] = py * SH * PW + px;
  }

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // weights: fragment order [N/16][K/32][64][8], K = tap*Cin + cin.  K-step index of (tap, chunk cc, ks):
  //   (tap*Cin + cc*64 + ks*32) / 32 = tap*(Cin/32) + cc*2 + ks
  const bf16_t* wfp = p.wfrag + (long long)lane * 8;
  long long wtile[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i) wtile[i] = (long long)min(n0 / 16 + wn * FN + i, p.n16 - 1) * p.k32 * 512;
  const int cin32 = p.Cin >> 5;
  bf16x8 wcur[FN], wnxt[FN];
  auto load_w = [&](int kidx, bf16x8 (&dst)[FN]) {
#pragma unroll
    for (int i = 0; i < FN; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(wfp + wtile[i] + (long long)kidx * 512);
  };

  load_patch(0);
  load_w(0, wcur);
  store_patch();
  __syncthreads();

  for (int cc = 0; cc < nch; ++cc) {
    if (cc + 1 < nch) load_patch(cc + 1);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {   // not unrolled: keeps the weight prefetch exactly one k-step deep
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // prefetch the next k-step's weight fragments (next ks, next tap, or tap 0 of the next chunk)
        {
          int ntap = tap, nks = ks + 1, ncc = cc;
          if (nks == 2) { nks = 0; ++ntap; }
          if (ntap == 9) { ntap = 0; ++ncc; }
          if (ncc < nch) load_w(ntap * cin32 + ncc * 2 + nks, wnxt);
        }
        bf16x8 xf[FM];
#pragma unroll
        for (int j = 0; j < FM; ++j) {
          const int pp = pbase[j] + dy * PW + dx;
          xf[j] = *reinterpret_cast<const bf16x8*>(smem + pp * 128 + (((ks * 4 + fq) ^ (pp & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
          for (int j = 0; j < FM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcur[i], xf[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < FN; ++i) wcur[i] = wnxt[i];
      }
    }
    if (cc + 1 < nch) {
      __syncthreads();      // every wave is done reading the patch of chunk cc
      store_patch();
      __syncthreads();
    }
  }

  // Plain bf16 store with bias, optional same-size residual and none / ReLU - what every layer of the detector body
  // and the recogniser's trunk asks for - takes a lean path: the general epilogue below decides its modes per value,
  // several hundred VALU instructions per tile and wave, as many issue cycles as the tile's MFMAs.
  if (p.out_mode == 0 && !p.y_f32 && p.res_mode <= 1 && (p.act == ACT_NONE || p.act == ACT_RELU) &&
      (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0) {
    bf16_t* yp = reinterpret_cast<bf16_t*>(p.y);
    const bool relu = p.act == ACT_RELU;
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n0 + wn * TN + i * 16 + fq * 4;
      if (n >= p.Cout) continue;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int lp = wm * TM + j * 16 + frow;
        const int py = lp / TW, px = lp - py * TW;
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy >= p.Ho || ox >= p.Wo) continue;
        const long long m = ((long long)b * p.Ho + oy) * p.Wo + ox;
        float v0 = acc[i][j][0] + bv[0], v1 = acc[i][j][1] + bv[1], v2 = acc[i][j][2] + bv[2], v3 = acc[i][j][3] + bv[3];
        if (p.res_mode == 1) {
          const u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + m * p.res_cstride + n);
          v0 += bf16_bits_to_f32(rr[0] & 0xffffu);
          v1 += bf16_bits_to_f32(rr[0] >> 16);
          v2 += bf16_bits_to_f32(rr[1] & 0xffffu);
          v3 += bf16_bits_to_f32(rr[1] >> 16);
        }
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        const bf16x4 o = {(bf16_t)v0, (bf16_t)v1, (bf16_t)v2, (bf16_t)v3};
        *reinterpret_cast<bf16x4*>(yp + m * p.y_cstride + p.y_coff + n) = o;
      }
    }
    return;
  }
  // Replicate store (the FPN's p3 / p4 / p5 written f x f times into the concat buffer at 1/4 scale): written straight
  // from the accumulators every copy is an 8-byte piece of a 128-byte pixel row - 64 partial-line writes per value for
  // p5.  The tile is put into LDS once (bias and activation applied, bf16) and every thread then copies 16-byte chunks,
  // eight consecutive lanes per destination pixel: whole 128-byte lines, half as many store instructions.
  if (p.out_mode == 2 && !p.y_f32 && p.res_mode == 0 && BN == 64 && ((p.y_cstride | p.y_coff | p.Cout) & 7) == 0) {
    constexpr int ROWB = 144;                       // 128 bytes of channels + 16: consecutive pixel rows start 4 banks apart
    static_assert(BM * ROWB <= PH * PW * 128, "the staged tile must fit the patch buffer");
    __syncthreads();                                // every wave is done reading the last patch
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int nl = wn * TN + i * 16 + fq * 4;
      const int n = n0 + nl;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && n < p.Cout) bv = f32x4{p.bias[n], p.bias[n + 1], p.bias[n + 2], p.bias[n + 3]};
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int lp = wm * TM + j * 16 + frow;
        const bf16x4 o = {(bf16_t)dlamd_act(acc[i][j][0] + bv[0], p.act), (bf16_t)dlamd_act(acc[i][j][1] + bv[1], p.act),
                          (bf16_t)dlamd_act(acc[i][j][2] + bv[2], p.act), (bf16_t)dlamd_act(acc[i][j][3] + bv[3], p.act)};
        *reinterpret_cast<bf16x4*>(smem + lp * ROWB + nl * 2) = o;
      }
    }
    __syncthreads();
    const int f = p.out_f;
    const int chunks = min(BN, p.Cout - n0) >> 3;   // valid 16-byte chunks of a pixel row
    bf16_t* yp = reinterpret_cast<bf16_t*>(p.y);
    const int total = BM * 8 * f * f;
    for (int q = tid; q < total; q += NT) {
      const int c = q & 7;
      const int rest = q >> 3;
      const int lp = rest % BM, rep = rest / BM;
      const int py = lp / TW, px = lp - py * TW;
      const int oy = oy0 + py, ox = ox0 + px;
      if (c >= chunks || oy >= p.Ho || ox >= p.Wo) continue;
      const int dy = rep / f, dx = rep - dy * f;
      const long long pix = ((long long)b * (f * p.Ho) + f * oy + dy) * (f * p.Wo) + f * ox + dx;
      *reinterpret_cast<u32x4*>(yp + pix * p.y_cstride + p.y_coff + n0 + c * 8) =
          *reinterpret_cast<const u32x4*>(smem + lp * ROWB + c * 16);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int lp = wm * TM + j * 16 + frow;
    const int py = lp / TW, px = lp - py * TW;
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy >= p.Ho || ox >= p.Wo) continue;
    const int m = (b * p.Ho + oy) * p.Wo + ox;
    conv_epilogue_pixel<FN, FM>(p, acc, j, m, n0 + wn * TN, fq);
  }
}

template <int TH, int TW, int SH, int BN, int WM, int WN>
static int launch_patch(ConvArgs& a, hipStream_t stream) {
  constexpr int PH = (TH - 1) * SH + 3, PW = TW + 2;
  const size_t smem = (size_t)PH * PW * 128;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const long long grid = (long long)a.B * ((a.Ho + TH - 1) / TH) * ((a.Wo + TW - 1) / TW) * a.tiles_n;
  if (grid <= 0 || grid > 0x7fffffffLL) return -3;
  auto kern = conv3x3_patch_kernel<TH, TW, SH, BN, WM, WN>;
  if (smem > 65536 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
    return -2;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), smem, stream, a);
  return DLAMD_CHECK_LAUNCH();
}

// 256-pixel tiles (8 waves as 4 x 2): half the weight traffic per output pixel of the 128-pixel tile
template <int SH, int BN>
static int dispatch_patch256(ConvArgs& a, hipStream_t stream) {
  int best = 16;
  double bw = 1e9;
  for (int th : {16, 8, 4, 2}) {
    const int tw = 256 / th;
    double w = (double)(((a.Ho + th - 1) / th) * th) / a.Ho * (double)(((a.Wo + tw - 1) / tw) * tw) / a.Wo *
               (1.0 + 2.0 / (th * SH)) * (1.0 + 2.0 / tw);
    if (w < bw - 1e-9) { bw = w; best = th; }
  }
  switch (best) {
    case 16: return launch_patch<16, 16, SH, BN, 4, 2>(a, stream);
    case 8: return launch_patch<8, 32, SH, BN, 4, 2>(a, stream);
    case 4: return launch_patch<4, 64, SH, BN, 4, 2>(a, stream);
    default: return launch_patch<2, 128, SH, BN, 4, 2>(a, stream);
  }
}

// Tile-shape choice for the patch kernel: the tallest tile that wastes the fewest rows of Ho.
template <int SH, int BN, int WM, int WN>
static int dispatch_patch(ConvArgs& a, hipStream_t stream) {
  auto waste = [&](int th) { return ((a.Ho + th - 1) / th) * th - a.Ho; };
  int best = 8;
  for (int th : {4, 2, 1})
    if (waste(th) * best < waste(best) * th || (waste(best) > 0 && waste(th) == 0)) best = th;
  // relative waste comparison above is crude; recompute exactly on padded-row fraction
  double bw = 1e9;
  for (int th : {8, 4, 2, 1}) {
    double w = (double)(((a.Ho + th - 1) / th) * th) / a.Ho * (1.0 + 2.0 / (th * SH));   // row padding x halo overhead
    if (w < bw - 1e-9) { bw = w; best = th; }
  }
  switch (best) {
    case 8: return launch_patch<8, 16, SH, BN, WM, WN>(a, stream);
    case 4: return launch_patch<4, 32, SH, BN, WM, WN>(a, stream);
    case 2: return launch_patch<2, 64, SH, BN, WM, WN>(a, stream);
    default: return launch_patch<1, 128, SH, BN, WM, WN>(a, stream);
  }
}

template <int BM, int BN, int WM, int WN, bool IS_1X1, bool B_DIRECT, int BKT>
static int launch_impl(ConvArgs& a, hipStream_t stream) {
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const int nkt = (a.K + BKT - 1) / BKT;
  // K <= 64 (1x1 laterals, deconvs): one K tile, no double buffer -> half the LDS, twice the blocks per CU.
  a.nbuf = nkt == 1 ? 1 : 2;
  size_t smem = a.nbuf * (BM + (B_DIRECT ? 0 : BN)) * (BKT * 2) + (IS_1X1 ? 0 : size_t(nkt) * (BKT / 8) * 2 * sizeof(uint32_t));
  if (a.coalesce && BM / WM / 16 <= 4) smem = smem > size_t(BM) * (BN * 2 + 16) ? smem : size_t(BM) * (BN * 2 + 16);
  auto kern = conv_igemm_kernel<BM, BN, WM, WN, IS_1X1, B_DIRECT, BKT>;
  static size_t max_set = 0;
  if (smem > 65536 && smem > max_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return -2;
    max_set = 160 * 1024;
  }
  long long grid = (long long)a.tiles_m * a.tiles_n;
  if (grid <= 0 || grid > 0x7fffffffLL) return -3;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, stream, a);
  return DLAMD_CHECK_LAUNCH();
}

template <int BM, int BN, int WM, int WN, bool IS_1X1>
static int launch_persistent(ConvArgs& a, hipStream_t stream) {
  if (!a.wfrag) return -14;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const int nkt = (a.K + 63) / 64;
  a.nbuf = nkt == 1 ? 1 : 2;
  size_t front = size_t(a.nbuf) * BM * 128;
  if (front < size_t(BM) * (BN * 2 + 16)) front = size_t(BM) * (BN * 2 + 16);      // operand buffers | output image
  const size_t smem = front + (IS_1X1 ? 0 : size_t(nkt) * 8 * 2 * sizeof(uint32_t));
  if (smem > 65536) return -17;
  auto kern = conv_igemm_persistent_kernel<BM, BN, WM, WN, IS_1X1>;
  long long tiles = (long long)a.tiles_m * a.tiles_n;
  if (tiles <= 0 || tiles > 0x7fffffffLL) return -3;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -2;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, smem) != hipSuccess || per_cu <= 0) per_cu = 2;
  long long slots = (long long)n_cu * per_cu;
  if (const char* e = getenv("DLAMD_CONV_PERSIST_GRID")) {      // tests: force several tiles per workgroup on small problems
    const long long v = atoll(e);
    if (v > 0 && v < slots) slots = v;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < slots ? tiles : slots)), dim3(256), smem, stream, a);
  return DLAMD_CHECK_LAUNCH();
}

template <int BM, int BN, int WM, int WN, bool IS_1X1>
static int launch_cfg(ConvArgs& a, hipStream_t stream) {
  if (a.wfrag) return launch_impl<BM, BN, WM, WN, IS_1X1, true, 64>(a, stream);
  return launch_impl<BM, BN, WM, WN, IS_1X1, false, 64>(a, stream);
}

template <bool IS_1X1>
static int dispatch(ConvArgs& a, hipStream_t stream, int cfg) {
  // cfg: 0 auto; otherwise explicit BN choice
  int bn = cfg;
  if (bn == 0) {
    if (a.Cout <= 16) bn = 16;
    else if (a.Cout <= 32) bn = 32;
    else if (a.Cout <= 64) bn = 64;
    else bn = 1128;   // 64 x 128 tile: with weights read directly (no LDS) it out-runs 128 x 128 on every wide layer
  }
  if (cfg == 0) {
    // Small-M GEMMs (decoder steps: M = batch) would launch a handful of 128-row tiles on 256 CUs and
    // run latency-bound; shrink the tile until the grid covers the chip (weights are then streamed by
    // many CUs in parallel, x stays L2-resident).
    const int bn_w = bn == 1128 ? 128 : bn;
    const long long t128 = (long long)((a.M + (bn == 1128 ? 63 : 127)) / (bn == 1128 ? 64 : 128)) * ((a.Cout + bn_w - 1) / bn_w);
    if (t128 < 160 && a.Cout >= 32) {
      const long long t64 = (long long)((a.M + 63) / 64) * ((a.Cout + 63) / 64);
      bn = (t64 >= 160 || a.Cout < 64) ? 1064 : 2032;
      if (a.Cout < 64) bn = 2032;
      // deep-K small-M GEMMs (decoder FFN down-projection, K = 4096): a 128-deep K tile halves the exposed load
      // latencies of this latency-bound shape: 32.4 -> 22.8 us at M 256 (bench/small_m_probe.py); K = 1024 ties
      if (bn == 2032 && a.K >= 2048 && a.wfrag) bn = 4032;
    }
  }
  // 1x1 layers (FPN laterals, deconv GEMMs, LSTM input projections, FC): the persistent form of the 64 x 128 tile
  // is 5-9 % faster (no tap table, stores drain under the next tile); on 3x3 layers its 128-register cap loses.
  if (cfg == 0 && bn == 1128 && IS_1X1 && a.wfrag) bn = 8128;
  switch (bn) {
    case 2032: return launch_cfg<32, 32, 2, 2, IS_1X1>(a, stream);
    case 16: return launch_cfg<128, 16, 4, 1, IS_1X1>(a, stream);
    case 32: return launch_cfg<128, 32, 4, 1, IS_1X1>(a, stream);
    case 64: return launch_cfg<128, 64, 2, 2, IS_1X1>(a, stream);
    case 128: return launch_cfg<128, 128, 2, 2, IS_1X1>(a, stream);
    case 1064: return launch_cfg<64, 64, 2, 2, IS_1X1>(a, stream);
    case 1128: return launch_cfg<64, 128, 1, 4, IS_1X1>(a, stream);
    case 4128:   // 64 x 128 tile with a 128-deep K tile (needs fragment weights); measured: ties or loses to 1128
                 // except on 512->512 (fewer barriers, but 132 VGPRs -> 3 waves/SIMD), so auto never picks it
      if (!a.wfrag) return -14;
      return launch_impl<64, 128, 1, 4, IS_1X1, true, 128>(a, stream);
    case 4032:   // 32 x 32 tile (small-M GEMMs: decoder steps), 128-deep K tile: half the exposed load latencies
      if (!a.wfrag) return -14;
      return launch_impl<32, 32, 2, 2, IS_1X1, true, 128>(a, stream);
    case 8128: return launch_persistent<64, 128, 1, 4, IS_1X1>(a, stream);   // persistent 64 x 128
    // Bigger REGISTER-staged tiles with direct weights (128x128, 256x128, 128x256 as 1x4 waves) were measured
    // twice and dropped (profiles/06_conv_tile_sweep.txt): first 4-5x slower because their unrolled epilogue put
    // the accumulators in scratch, then - scratch-free, rolled epilogue - still 1.25x / 2x / 1.6x slower than
    // 64x128: their accumulators (68-132 AGPRs) cut the resident workgroups this loop needs to hide latency.
    default: return -4;
  }
}

// conv_stem.hip: 3x3 / pad 1 / small-Cin first-layer kernels
bool dlamd_conv_stem_serves(int Cin, int stride, int Cout);
int dlamd_conv_stem(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int stride, int Ho,
                    int Wo, int Cout, int act, int y_cstride, int y_coff, hipStream_t stream);

// conv_lateral.hip: streaming 1x1 kernel for Cin 64 / 128 -> 256 channels (FPN laterals)
bool dlamd_conv_lateral_serves(int Cin, int x_cstride, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                               int res_cstride, int res_mode, int out_mode, int y_cstride, int y_coff, int y_f32);
int dlamd_conv_lateral(const void* x, const void* w, const float* bias, const void* res, void* y, int M, int H, int W,
                       int Cin, int x_cstride, int act, int res_mode, int res_cstride, int y_cstride, int y_coff,
                       hipStream_t stream);

extern "C" int dlamd_conv2d(const void* x, const void* w, const float* bias, const void* res, void* y, void* mask,
                            int B, int H, int W, int Cin, int x_cstride, int Cout, int kh, int kw, int sh, int sw,
                            int ph, int pw, int act, int res_mode, int res_cstride, int out_mode, int out_f,
                            int y_cstride, int y_coff, int y_f32, float thr, int cfg, const void* wfrag,
                            int wfrag_n16, int wfrag_k32, hipStream_t stream) {
  ConvArgs a;
  a.wfrag = (const bf16_t*)wfrag;
  a.k32 = (kh * kw * Cin + 127) / 128 * 4;   // weight fragments are zero-padded to a multiple of 128 in K
  a.n16 = wfrag_n16;
  // host-side contract for the fragment buffer: the caller states its real extent ([n16][k32][64][8]); the K
  // extent must be the padded K the kernels step through and the N extent must cover every output channel
  if (wfrag && (wfrag_k32 != a.k32 || wfrag_n16 < (Cout + 15) / 16)) return -15;
  a.x = (const bf16_t*)x;
  a.w = (const bf16_t*)w;
  a.bias = bias;
  a.res = (const bf16_t*)res;
  a.y = y;
  a.mask = (uint8_t*)mask;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.x_cstride = x_cstride;
  a.Ho = (H + 2 * ph - kh) / sh + 1;
  a.Wo = (W + 2 * pw - kw) / sw + 1;
  a.Cout = Cout;
  a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw;
  a.K = kh * kw * Cin;
  long long M = (long long)B * a.Ho * a.Wo;
  if (M <= 0 || M > 0x7fffffffLL) return -5;
  a.M = (int)M;
  a.act = act;
  a.res_mode = res ? res_mode : 0;
  a.res_cstride = res_cstride;
  a.out_mode = out_mode;
  a.out_f = out_f;
  a.y_cstride = y_cstride;
  a.y_coff = y_coff;
  a.y_f32 = y_f32;
  a.thr = thr;
  a.coalesce = (out_mode == 0 && !y_f32 && y_cstride % 8 == 0 && y_coff % 8 == 0) ? 1 : 0;
  // Host-side shape contract of the kernel (16-byte vector loads / 8-byte stores).
  if (Cin % 8 != 0 || x_cstride % 8 != 0 || a.K % 8 != 0) return -6;
  if (out_mode == 3) { if (Cout != 4 || mask == nullptr) return -7; }
  else if (Cout % 4 != 0 || y_cstride % 4 != 0 || y_coff % 4 != 0) return -8;
  if (out_mode == 1 && ((Cout / 4) % 4 != 0)) return -9;
  if (a.res_mode != 0 && res_cstride % 4 != 0) return -10;
  if (a.Ho <= 0 || a.Wo <= 0) return -11;
  if ((long long)a.K > 32768) return -12;
  // First layers (Cin 8 stride 2; Cin 32 stride 1 / 2): memory-bound stencil kernels (conv_stem.hip)
  // (cfg 9008 forces it, any other explicit cfg keeps the generic kernels for tests / sweeps)
  const bool stem_ok = x_cstride == Cin && kh == 3 && kw == 3 && sh == sw && ph == 1 && pw == 1 && out_mode == 0 &&
                       a.res_mode == 0 && !y_f32 && dlamd_conv_stem_serves(Cin, sh, Cout) && y_cstride % 8 == 0 &&
                       y_coff % 8 == 0;
  if (stem_ok && (cfg == 0 || cfg == 9008))
    return dlamd_conv_stem(x, w, bias, y, B, H, W, Cin, sh, a.Ho, a.Wo, Cout, act, y_cstride, y_coff, stream);
  if (cfg == 9008) return -13;
  // Shallow-K 1x1 into 256 channels over a large map (FPN lateral of the quarter-scale level): weights resident in
  // LDS, pixels streamed (cfg 9100 forces it; auto mode from 32 k pixels up, below that the prologue shows)
  const bool lat_ok = dlamd_conv_lateral_serves(Cin, x_cstride, Cout, kh, kw, sh, sw, ph, pw, res_cstride, a.res_mode,
                                                out_mode, y_cstride, y_coff, y_f32);
  if (lat_ok && ((cfg == 0 && a.M >= 32768) || cfg == 9100))
    return dlamd_conv_lateral(x, w, bias, a.res_mode ? res : nullptr, y, a.M, a.Ho, a.Wo, Cin, x_cstride, act,
                              a.res_mode, res_cstride, y_cstride, y_coff, stream);
  if (cfg == 9100) return -13;
  // cfg 3xxx forces / 39xx forbids the LDS-patch 3x3 kernel (benchmarks and tests); default: use it when it applies
  const bool patch_ok = a.wfrag && kh == 3 && kw == 3 && ph == 1 && pw == 1 && sw == 1 && (sh == 1 || sh == 2) &&
                        Cin % 64 == 0 && x_cstride == Cin && (long long)B * a.Ho * a.Wo < 0x7fffffffLL;
  // Measured (bench/kernel_bench.py --sweep, profiles/06_conv_tile_sweep.txt): the patch kernel wins by 1.3-1.5x when
  // Cout <= 64 (the im2col re-gather dominates: DBNet FPN smooth convs, head, layer1) and ties / loses against
  // the 64x128 direct-weight tile on wide layers, so auto mode uses it for narrow outputs only.
  if (patch_ok && cfg == 3264)      // 64-channel tile as 2 x 2 waves: every wave streams half of the weights
    return sh == 1 ? dispatch_patch<1, 64, 2, 2>(a, stream) : dispatch_patch<2, 64, 2, 2>(a, stream);
  if (patch_ok && cfg == 3464)      // 256 pixels x 64 channels, 8 waves
    return sh == 1 ? dispatch_patch256<1, 64>(a, stream) : dispatch_patch256<2, 64>(a, stream);
  // (A 512-pixel tile, 16 x 32 with 8 waves, halves the weight traffic once more but leaves one workgroup per CU: layer1
  // 0.268 ms against 0.227, head 0.710 against 0.652 - measured and dropped.)
  if (patch_ok && cfg == 3428)      // 256 pixels x 128 channels, 8 waves
    return sh == 1 ? dispatch_patch256<1, 128>(a, stream) : dispatch_patch256<2, 128>(a, stream);
  // Auto mode, stride 1 (bench/conv64_probe.py, 25 detector pages): the 64-channel tile as 2 x 2 waves, where every
  // wave streams half of the weights from L2 instead of all of them, beats the 4 x 1 arrangement by 15-25 % on every
  // Cout <= 64 layer (layer1 0.375 -> 0.290 ms, head 0.82 -> 0.68, FPN smooth 0.216 -> 0.182) and the 64 x 128
  // direct-weight tile by 5-7 % on the 128-channel layers with shallow K; on the quarter-scale maps the 256-pixel
  // 8-wave tile halves the weight traffic once more (layer1 0.270 ms, head 0.667).
  if (patch_ok && cfg == 0 && sh == 1 && (Cout <= 64 || (Cout <= 128 && Cin <= 128))) {
    if (Cout <= 64 && (long long)B * a.Ho * a.Wo >= 1000000 && a.Ho >= 16) return dispatch_patch256<1, 64>(a, stream);
    return dispatch_patch<1, 64, 2, 2>(a, stream);
  }
  if (patch_ok && ((cfg == 0 && Cout <= 64 && sh == 1) || cfg == 3064 || cfg == 3128)) {
    const bool wide = cfg == 3128 || (cfg == 0 && Cout > 64);
    if (sh == 1) return wide ? dispatch_patch<1, 128, 2, 2>(a, stream) : dispatch_patch<1, 64, 4, 1>(a, stream);
    return wide ? dispatch_patch<2, 128, 2, 2>(a, stream) : dispatch_patch<2, 64, 4, 1>(a, stream);
  }
  if (cfg == 3064 || cfg == 3128) return -13;
  if (cfg == 3900) cfg = 0;
  const bool is1x1 = (kh == 1 && kw == 1 && ph == 0 && pw == 0);
  return is1x1 ? dispatch<true>(a, stream, cfg) : dispatch<false>(a, stream, cfg);
}
