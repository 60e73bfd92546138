// 1x1 convolution with a shallow K (Cin 64 / 128) and 256 output channels over a large map: the FPN lateral of
// the detector's quarter-scale level (2 M pixels x 64 -> 256 per 25-page batch, + the nearest-2x upsampled
// coarser level as residual).  The layer moves 1.3 GB for 66 GFLOP, so it is a streaming kernel, not a GEMM tile:
//
//  * the whole weight matrix sits in LDS in MFMA fragment order for the life of the workgroup (32 / 64 KB) - no
//    per-tile weight stream from L2, no barrier after the prologue;
//  * every wave owns 16 pixels per step: one 16-byte load per lane and K-step brings them in as the B operand,
//    16 channel tiles x K/32 MFMAs produce all 256 channels of those pixels in 64 accumulator registers;
//  * the input of the next step is in flight while the current one is multiplied and stored;
//  * bias is the accumulator's initial value, the residual is added in fp32 at fragment granularity (8-byte
//    reads that hit L1 / L2: each coarse pixel is read by four output pixels), and the bf16 result goes through a
//    per-wave LDS slab so that global stores are 16 bytes per lane over whole 256-byte runs of a pixel row;
//  * a workgroup walks a contiguous range of pixels, so the two output rows that share a residual row are
//    handled by the same XCD (private L2).
#include "common.h"

struct LateralArgs {
  const bf16_t* x;
  const bf16_t* w;      // [256][K]
  const float* bias;    // [256] or nullptr
  const bf16_t* res;    // residual or nullptr
  bf16_t* y;
  int M, H, W;          // output pixels; map height / width (residual addressing)
  int x_cstride, res_mode, res_cstride, y_cstride, y_coff, act;
  int ntiles, chunk;    // tiles of NW * 16 pixels; tiles per workgroup
};

static constexpr int LAT_N = 256;          // output channels
static constexpr int LAT_SLAB_CH = 128;    // channels per store pass
static constexpr int LAT_SLAB_ROW = LAT_SLAB_CH * 2 + 16;   // bytes per pixel row of the slab (+16: bank spread)

template <int KS, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(KS == 2 ? 4 : 2, KS == 2 ? 4 : 2))) void lateral1x1_kernel(const LateralArgs p) {
  constexpr int K = KS * 32;
  extern __shared__ __align__(16) char lat_smem[];
  u32x4* wl = reinterpret_cast<u32x4*>(lat_smem);                        // [16 tiles][KS][64 lanes]
  float* bl = reinterpret_cast<float*>(lat_smem + 16 * KS * 1024);       // [256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  char* slab = lat_smem + 16 * KS * 1024 + LAT_N * 4 + wave * 16 * LAT_SLAB_ROW;
  const int px = lane & 15, kq = lane >> 4;

  for (int e = tid; e < 16 * KS * 64; e += NW * 64) {
    const int l = e & 63, s = (e >> 6) % KS, t = (e >> 6) / KS;
    wl[e] = *reinterpret_cast<const u32x4*>(p.w + (size_t)(t * 16 + (l & 15)) * K + s * 32 + (l >> 4) * 8);
  }
  for (int e = tid; e < LAT_N; e += NW * 64) bl[e] = p.bias ? p.bias[e] : 0.f;
  __syncthreads();

  const int t0 = blockIdx.x * p.chunk;
  const int t1 = min(t0 + p.chunk, p.ntiles);
  u32x4 xb[KS];
  auto load_x = [&](int tile) {
    const int m = (tile * NW + wave) * 16 + px;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      xb[s] = u32x4{0u, 0u, 0u, 0u};
      if (m < p.M) xb[s] = *reinterpret_cast<const u32x4*>(p.x + (size_t)m * p.x_cstride + s * 32 + kq * 8);
    }
  };
  if (t0 < t1) load_x(t0);
  const int hw = p.H * p.W;
  const int Hr = (p.H + 1) >> 1, Wr = (p.W + 1) >> 1;

  for (int tile = t0; tile < t1; ++tile) {
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = *reinterpret_cast<const f32x4*>(bl + t * 16 + kq * 4);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bf16x8 b = as_bf16x8(xb[s]);
#pragma unroll
      for (int t = 0; t < 16; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wl[(t * KS + s) * 64 + lane]), b, acc[t], 0, 0, 0);
    }
    const int m0 = (tile * NW + wave) * 16;
    if (tile + 1 < t1) load_x(tile + 1);

    const int m = m0 + px;
    const bf16_t* rr = nullptr;
    if (p.res_mode != 0 && m < p.M) {
      long long rp = m;
      if (p.res_mode == 2) {
        const int b = m / hw, rem = m - b * hw;
        const int ho = rem / p.W, wo = rem - ho * p.W;
        rp = ((long long)b * Hr + (ho >> 1)) * Wr + (wo >> 1);
      }
      rr = p.res + rp * p.res_cstride + kq * 4;
    }

    // The slab belongs to this wave alone: LDS operations of one wave complete in issue order, so a fence that
    // keeps the compiler from moving them is all the synchronisation the write -> read-back -> rewrite needs.
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (rr) {
        u32x2 r[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) r[t] = *reinterpret_cast<const u32x2*>(rr + (half * 8 + t) * 16);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          f32x4& a = acc[half * 8 + t];
          a[0] += bf16_bits_to_f32(r[t][0] & 0xffffu);
          a[1] += bf16_bits_to_f32(r[t][0] >> 16);
          a[2] += bf16_bits_to_f32(r[t][1] & 0xffffu);
          a[3] += bf16_bits_to_f32(r[t][1] >> 16);
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f32x4 v = acc[half * 8 + t];
        if (p.act != ACT_NONE) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = dlamd_act(v[q], p.act);
        }
        const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(slab + px * LAT_SLAB_ROW + (t * 16 + kq * 4) * 2) = o;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = i * 64 + lane, row = c >> 4, cc = c & 15;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab + row * LAT_SLAB_ROW + cc * 16);
        if (m0 + row < p.M)
          *reinterpret_cast<u32x4*>(p.y + (size_t)(m0 + row) * p.y_cstride + p.y_coff + half * LAT_SLAB_CH + cc * 8) = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

bool dlamd_conv_lateral_serves(int Cin, int x_cstride, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                               int res_cstride, int res_mode, int out_mode, int y_cstride, int y_coff, int y_f32) {
  return (Cin == 64 || Cin == 128) && x_cstride % 8 == 0 && Cout == LAT_N && kh == 1 && kw == 1 && sh == 1 && sw == 1 &&
         ph == 0 && pw == 0 && out_mode == 0 && !y_f32 && y_cstride % 8 == 0 && y_coff % 8 == 0 &&
         (res_mode == 0 || (res_cstride % 4 == 0 && res_cstride >= LAT_N));
}

template <int KS>
static int launch_lateral(LateralArgs a, hipStream_t stream) {
  constexpr int NW = 8;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    cus = prop.multiProcessorCount;
  }
  const size_t smem = 16 * KS * 1024 + LAT_N * 4 + NW * 16 * LAT_SLAB_ROW;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(lateral1x1_kernel<KS, NW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return -1;
    attr_set = true;
  }
  a.ntiles = (a.M + NW * 16 - 1) / (NW * 16);
  const int per_cu = smem <= 80 * 1024 ? 2 : 1;
  int grid = min(a.ntiles, cus * per_cu);
  a.chunk = (a.ntiles + grid - 1) / grid;
  grid = (a.ntiles + a.chunk - 1) / a.chunk;
  hipLaunchKernelGGL((lateral1x1_kernel<KS, NW>), dim3(grid), dim3(NW * 64), smem, stream, a);
  return DLAMD_CHECK_LAUNCH();
}

int dlamd_conv_lateral(const void* x, const void* w, const float* bias, const void* res, void* y, int M, int H, int W,
                       int Cin, int x_cstride, int act, int res_mode, int res_cstride, int y_cstride, int y_coff,
                       hipStream_t stream) {
  LateralArgs a;
  a.x = (const bf16_t*)x;
  a.w = (const bf16_t*)w;
  a.bias = bias;
  a.res = (const bf16_t*)res;
  a.y = (bf16_t*)y;
  a.M = M; a.H = H; a.W = W;
  a.x_cstride = x_cstride;
  a.res_mode = res ? res_mode : 0;
  a.res_cstride = res_cstride;
  a.y_cstride = y_cstride;
  a.y_coff = y_coff;
  a.act = act;
  a.ntiles = 0; a.chunk = 0;
  return Cin == 64 ? launch_lateral<2>(a, stream) : launch_lateral<4>(a, stream);
}
