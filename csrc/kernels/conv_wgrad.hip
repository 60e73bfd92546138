// Weight gradient of an NHWC convolution (training):
//   dW[co][r][s][ci] = sum over output pixels p of dY[p][co] * X[q(p, r, s)][ci]
// As a GEMM per tap the reduction runs over PIXELS, and both operands are stored channel-contiguous - the
// wrong way round for MFMA fragments (a lane needs 8 consecutive reduction indices of one row / column).  The tiles are
// therefore staged in LDS as they are in memory ([pixel][channel]) and read back with the gfx950 transposed LDS read
// (`ds_read_b64_tr_b16`: a 16-lane group fetches 4 rows x 16 columns and every lane receives one column).
//
// A workgroup owns dW[64 output channels][all taps][64 input channels] (up to 9 taps: 36 accumulator tiles per wave)
// and walks over units of 64 consecutive output pixels of one output row (two MFMA K-steps per barrier pair):
//   * the dY tile [64][64] and the input patch under it ([kh][63 * sw + kw][64], zero outside the image) are fetched
//     into registers one unit ahead and written to LDS after the previous unit's MFMAs;
//   * wave w takes input channels 16 w .. 16 w + 15: per K-step it reads 4 dY fragments (one per 16 output channels) and
//     one patch fragment per tap - 13 fragments for 36 MFMAs, so the LDS port stays below the MFMA time;
//   * units are spread over gridDim.x workgroups (split-K); partial sums are added to dW with float atomics, so the
//     caller zero-fills dW first.
// Shape contract (checked on the host): Cin % 64 == 0, Cout % 64 == 0, kh * kw <= 9, x / dy pixel-contiguous.
#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
  const bf16_t* x;     // [N, H, W, Cin]
  const bf16_t* dy;    // [N, Ho, Wo, Cout]
  float* dw;           // [Cout, kh, kw, Cin], accumulated into
  int N, H, W, Cin, Ho, Wo, Cout;
  int kh, kw, sh, sw, ph, pw;
  int segs;            // ceil(Wo / WG_PX)
  long long units;     // N * Ho * segs
};

static constexpr int WG_PX = 64;            // output pixels per unit (two MFMA K-steps per barrier pair)
static constexpr int WG_PITCH = 64 * 2 + 16;  // bytes per pixel row of an LDS tile (64 channels + pad: bank spread)
static constexpr int WG_MAX_PW = (WG_PX - 1) * 2 + 3;  // widest patch row (stride 2, 3 taps)
static constexpr int WG_MAX_LOADS = (3 * WG_MAX_PW * 8 + 255) / 256;   // 16-byte patch chunks per thread

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row0, int row_step, int col0, int lane) {
  // rows row0 + row_step * (8 g + 4 h + q), columns col0 .. col0 + 15 -> MFMA operand fragment of lane group g
  const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const char* a0 = tile + (row0 + row_step * (8 * g + q)) * WG_PITCH + (col0 + 4 * p) * 2;
  const char* a1 = a0 + row_step * 4 * WG_PITCH;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
  __shared__ __align__(16) char dys[WG_PX * WG_PITCH];
  __shared__ __align__(16) char xs[3 * WG_MAX_PW * WG_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int taps = p.kh * p.kw;
  const int PW = (WG_PX - 1) * p.sw + p.kw;        // patch row length in pixels
  const int nload = (p.kh * PW * 8 + 255) / 256;   // 16-byte patch chunks per thread
  const u32x4 zero = {0u, 0u, 0u, 0u};

  f32x4 acc[4][9];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int DY_LOADS = WG_PX * 8 / 256;
  u32x4 rdy[DY_LOADS], rx[WG_MAX_LOADS];
  auto fetch = [&](long long u) {
    const int seg = (int)(u % p.segs);
    const long long r = u / p.segs;
    const int oy = (int)(r % p.Ho), n = (int)(r / p.Ho);
    const int ox0 = seg * WG_PX;
#pragma unroll
    for (int i = 0; i < DY_LOADS; ++i) {   // dY tile: pixel e / 8, 16-byte chunk e % 8
      const int e = i * 256 + tid, px = e >> 3, ch = e & 7, ox = ox0 + px;
      rdy[i] = zero;
      if (ox < p.Wo)
        rdy[i] = *reinterpret_cast<const u32x4*>(p.dy + (((long long)n * p.Ho + oy) * p.Wo + ox) * p.Cout + co0 + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < WG_MAX_LOADS; ++i) {
      const int e = i * 256 + tid;
      rx[i] = zero;
      if (i < nload && e < p.kh * PW * 8) {
        const int ch = e & 7, pix = e >> 3;
        const int r2 = pix / PW, c = pix - r2 * PW;
        const int iy = oy * p.sh + r2 - p.ph, ix = ox0 * p.sw + c - p.pw;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
          rx[i] = *reinterpret_cast<const u32x4*>(p.x + (((long long)n * p.H + iy) * p.W + ix) * p.Cin + ci0 + ch * 8);
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < DY_LOADS; ++i) {
      const int e = i * 256 + tid;
      *reinterpret_cast<u32x4*>(dys + (e >> 3) * WG_PITCH + (e & 7) * 16) = rdy[i];
    }
#pragma unroll
    for (int i = 0; i < WG_MAX_LOADS; ++i) {
      const int e = i * 256 + tid;
      if (i < nload && e < p.kh * PW * 8) *reinterpret_cast<u32x4*>(xs + (e >> 3) * WG_PITCH + (e & 7) * 16) = rx[i];
    }
  };

  long long u = blockIdx.x;
  if (u < p.units) fetch(u);
  while (u < p.units) {
    __syncthreads();               // the previous unit's fragment reads are done
    stash();
    __syncthreads();
    const long long un = u + gridDim.x;
    if (un < p.units) fetch(un);   // in flight under the MFMAs below
#pragma unroll
    for (int ks = 0; ks < WG_PX / 32; ++ks) {
      // all fragments of the K-step first (13 transposed reads in flight together), then its 36 MFMAs: with one wave
      // per SIMD nothing else hides an LDS round trip per tap
      bf16x8 a[4], b[9];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = tr_frag(dys, 32 * ks, 1, i * 16, lane);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // every lane executes the transposed read (it needs the whole wave); taps beyond kh * kw read tap 0 and are dropped
        const int tt = t < taps ? t : 0;
        const int r = tt / p.kw, s = tt - r * p.kw;
        b[t] = tr_frag(xs, r * PW + s + 32 * ks * p.sw, p.sw, wave * 16, lane);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < taps) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[t], acc[i][t], 0, 0, 0);
        }
      }
    }
    u = un;
  }

  // D: lane holds rows (output channels) 4 * (lane >> 4) + j of tile i, column (input channel) lane & 15 of this wave
  const int ci = ci0 + wave * 16 + (lane & 15);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t >= taps) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = co0 + i * 16 + 4 * (lane >> 4) + j;
        unsafeAtomicAdd(p.dw + ((long long)co * taps + t) * p.Cin + ci, acc[i][t][j]);
      }
    }
}

// x: [N,H,W,Cin] bf16, dy: [N,Ho,Wo,Cout] bf16 (both contiguous), dw: [Cout,kh,kw,Cin] fp32, ADDED to.
extern "C" int dlamd_conv_wgrad(const void* x, const void* dy, void* dw, int N, int H, int W, int Cin, int Cout, int kh,
                                int kw, int sh, int sw, int ph, int pw, hipStream_t stream) {
  if (!x || !dy || !dw || N <= 0 || H <= 0 || W <= 0) return -1;
  if (Cin % 64 != 0 || Cout % 64 != 0 || kh < 1 || kw < 1 || kh > 3 || kw > 3 || kh * kw > 9) return -2;
  if (sh < 1 || sw < 1 || sw > 2 || sh > 4 || ph < 0 || pw < 0 || ph >= kh + 1 || pw >= kw + 1) return -3;
  WgradArgs a;
  a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = (float*)dw;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.kh = kh; a.kw = kw; a.sh = sh; a.sw = sw; a.ph = ph; a.pw = pw;
  a.Ho = (H + 2 * ph - kh) / sh + 1;
  a.Wo = (W + 2 * pw - kw) / sw + 1;
  if (a.Ho <= 0 || a.Wo <= 0) return -4;
  a.segs = (a.Wo + WG_PX - 1) / WG_PX;
  a.units = (long long)N * a.Ho * a.segs;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -5;
    cus = prop.multiProcessorCount;
  }
  const int tiles = (Cout / 64) * (Cin / 64);
  long long gx = (cus + tiles - 1) / tiles;            // one 396-register workgroup fits a CU at a time
  if (gx > a.units) gx = a.units;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, Cout / 64, Cin / 64);
  hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), 0, stream, a);
  return DLAMD_CHECK_LAUNCH();
}
