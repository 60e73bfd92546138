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
};

static constexpr int BK = 64;  // K elements per tile (128 B per LDS row)

template <int BM, int BN, int WM, int WN, bool IS_1X1>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int PA = (BM + 31) / 32, PB = (BN + 31) / 32;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TM % 16 == 0 && TN % 16 == 0, "tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                       // 2 x BM x 128 B
  char* Bs = smem + 2 * BM * 128;        // 2 x BN x 128 B
  uint32_t* ktab = reinterpret_cast<uint32_t*>(smem + 2 * (BM + BN) * 128);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;

  const int nwg = gridDim.x;
  const int tile = dlamd_xcd_remap(blockIdx.x, nwg);
  const int tn = tile % p.tiles_n;
  const int tm = tile / p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = (p.K + BK - 1) / BK;

  const int c = tid & 7;        // 16-byte chunk inside the 128-byte K row
  const int r0 = tid >> 3;      // 0..31

  if (!IS_1X1) {
    // (tap, cin) decomposition of every 8-wide K chunk, once per block.
    for (int e = tid; e < nkt * 8; e += 256) {
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

  // Per-thread row descriptors (fixed across the K loop).
  long long a_base[PA];
  int a_hi0[PA], a_wi0[PA];
  bool a_ok[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int r = r0 + 32 * i;
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
    int r = r0 + 32 * i;
    int n = n0 + r;
    b_ok[i] = (r < BN) && (n < p.Cout);
    b_base[i] = (long long)(b_ok[i] ? n : 0) * p.K;
  }

  u32x4 a_reg[PA], b_reg[PB];

  auto load_tile = [&](int kt) {
    int k = kt * BK + c * 8;
    bool kok;
    int dy = 0, dx = 0;
    long long off;
    if (IS_1X1) {
      kok = k < p.K;
      off = k;
    } else {
      uint32_t t = ktab[2 * (kt * 8 + c)];
      kok = t != 0xFFFFFFFFu;
      dy = int(t >> 8);
      dx = int(t & 0xff);
      off = (long long)ktab[2 * (kt * 8 + c) + 1];
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
    bool kwok = k < p.K;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      u32x4 val = {0u, 0u, 0u, 0u};
      if (b_ok[i] && kwok) val = *reinterpret_cast<const u32x4*>(p.w + b_base[i] + k);
      b_reg[i] = val;
    }
  };

  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int r = r0 + 32 * i;
      if (BM % 32 == 0 || r < BM)
        *reinterpret_cast<u32x4*>(As + buf * BM * 128 + r * 128 + ((c ^ (r & 7)) << 4)) = a_reg[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int r = r0 + 32 * i;
      if (BN % 32 == 0 || r < BN)
        *reinterpret_cast<u32x4*>(Bs + buf * BN * 128 + r * 128 + ((c ^ (r & 7)) << 4)) = b_reg[i];
    }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const int fq = lane >> 4;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const char* Ab = As + buf * BM * 128;
    const char* Bb = Bs + buf * BN * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[FN], xf[FM];
      const int chunk = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        int row = wn * TN + i * 16 + frow;
        wf[i] = *reinterpret_cast<const bf16x8*>(Bb + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        int row = wm * TM + j * 16 + frow;
        xf[j] = *reinterpret_cast<const bf16x8*>(Ab + row * 128 + ((chunk ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  // acc[i][j][q]: channel n0 + wn*TN + i*16 + fq*4 + q, pixel m0 + wm*TM + j*16 + frow
  const int hw = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m0 + wm * TM + j * 16 + frow;
    if (m >= p.M) continue;
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
      const int n = n0 + wn * TN + i * 16 + fq * 4;
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
        *reinterpret_cast<bf16x4*>(yp + (long long)m * p.y_cstride + p.y_coff + n) = o;
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
}

template <int BM, int BN, int WM, int WN, bool IS_1X1>
static int launch_cfg(ConvArgs& a, hipStream_t stream) {
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (a.Cout + BN - 1) / BN;
  const int nkt = (a.K + BK - 1) / BK;
  size_t smem = 2 * (BM + BN) * 128 + (IS_1X1 ? 0 : size_t(nkt) * 8 * 2 * sizeof(uint32_t));
  auto kern = conv_igemm_kernel<BM, BN, WM, WN, IS_1X1>;
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

template <bool IS_1X1>
static int dispatch(ConvArgs& a, hipStream_t stream, int cfg) {
  // cfg: 0 auto; otherwise explicit BN choice
  int bn = cfg;
  if (bn == 0) {
    if (a.Cout <= 16) bn = 16;
    else if (a.Cout <= 32) bn = 32;
    else if (a.Cout <= 64) bn = 64;
    else bn = 128;
  }
  if (cfg == 0) {
    // Small-M GEMMs (decoder steps: M = batch) would launch a handful of 128-row tiles on 256 CUs and
    // run latency-bound; shrink the tile until the grid covers the chip (weights are then streamed by
    // many CUs in parallel, x stays L2-resident).
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.Cout + bn - 1) / bn);
    if (t128 < 160 && a.Cout >= 32) {
      const long long t64 = (long long)((a.M + 63) / 64) * ((a.Cout + 63) / 64);
      bn = (t64 >= 160 || a.Cout < 64) ? 1064 : 2032;
      if (a.Cout < 64) bn = 2032;
    }
  }
  switch (bn) {
    case 2032: return launch_cfg<32, 32, 2, 2, IS_1X1>(a, stream);
    case 16: return launch_cfg<128, 16, 4, 1, IS_1X1>(a, stream);
    case 32: return launch_cfg<128, 32, 4, 1, IS_1X1>(a, stream);
    case 64: return launch_cfg<128, 64, 2, 2, IS_1X1>(a, stream);
    case 128: return launch_cfg<128, 128, 2, 2, IS_1X1>(a, stream);
    case 1064: return launch_cfg<64, 64, 2, 2, IS_1X1>(a, stream);
    case 1128: return launch_cfg<64, 128, 1, 4, IS_1X1>(a, stream);
    default: return -4;
  }
}

extern "C" int dlamd_conv2d(const void* x, const void* w, const float* bias, const void* res, void* y, void* mask,
                            int B, int H, int W, int Cin, int x_cstride, int Cout, int kh, int kw, int sh, int sw,
                            int ph, int pw, int act, int res_mode, int res_cstride, int out_mode, int out_f,
                            int y_cstride, int y_coff, int y_f32, float thr, int cfg, hipStream_t stream) {
  ConvArgs a;
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
  // Host-side shape contract of the kernel (16-byte vector loads / 8-byte stores).
  if (Cin % 8 != 0 || x_cstride % 8 != 0 || a.K % 8 != 0) return -6;
  if (out_mode == 3) { if (Cout != 4 || mask == nullptr) return -7; }
  else if (Cout % 4 != 0 || y_cstride % 4 != 0 || y_coff % 4 != 0) return -8;
  if (out_mode == 1 && ((Cout / 4) % 4 != 0)) return -9;
  if (a.res_mode != 0 && res_cstride % 4 != 0) return -10;
  if (a.Ho <= 0 || a.Wo <= 0) return -11;
  if ((long long)a.K > 32768) return -12;
  const bool is1x1 = (kh == 1 && kw == 1 && ph == 0 && pw == 0);
  return is1x1 ? dispatch<true>(a, stream, cfg) : dispatch<false>(a, stream, cfg);
}
