// First-layer convolution: 3x3, stride 2, pad 1, Cin = 8 (RGB padded to 8 channels = one 16-byte pixel), Cout 32 / 64.
//
// The generic implicit-GEMM kernel gathers an im2col tile per K-tile; with K = 72 that is 9 x 16-byte gathers per
// output pixel for 2 MFMA K-steps of work, and the CRNN's first conv (8 M output pixels per 25-page batch) ran at
// 9 - 36 TFLOP/s, 20 % of all convolution time of the OCR bench (bench/conv_layer_breakdown.py, profiles/17).
// Here the layer is treated as what it is, a memory-bound stencil:
//   * a workgroup stages the input patch of an 8 x 64 output tile in LDS once (17 x 129 pixels, 35 KB), even and odd
//     input columns in separate planes so that the 16 pixels of an MFMA column group are contiguous for every tap;
//   * with K ordered (tap, channel) one MFMA B fragment row-group IS one pixel: lane (col, fq) of K-step s reads the
//     16-byte pixel of tap 4s+fq straight from the patch (ds_read_b128, conflict free) - no im2col copy at all;
//   * the whole weight tensor (Cout x 72, zero-padded to K = 96) lives in registers as A fragments;
//   * each wave transposes its 16-pixel x Cout results through a private LDS slab and stores whole 16-byte chunks.
// Same operand orientation as conv_igemm.hip (weights = A, pixels = B: a lane ends up with 4 consecutive channels).
#include "common.h"

namespace {

constexpr int ST_TH = 8, ST_TW = 64;               // output tile
constexpr int ST_PH = 2 * ST_TH + 1;               // 17 patch rows
constexpr int ST_PW = 2 * ST_TW + 1;               // 129 patch columns: 65 even + 64 odd
constexpr int ST_EVEN = ST_TW + 1;
constexpr int ST_PATCH_BYTES = ST_PH * ST_PW * 16;

struct StemArgs {
  const bf16_t* x;      // [B, H, W, 8]
  const bf16_t* w;      // [Cout, 3, 3, 8]
  const float* bias;    // [Cout] or nullptr
  bf16_t* y;            // [B, Ho, Wo, y_cstride] written at channel offset y_coff
  int B, H, W, Ho, Wo;
  int act, y_cstride, y_coff;
  int tiles_x, tiles_y;
};

template <int COUT>
__global__ __launch_bounds__(256) void conv_stem_c8s2_kernel(StemArgs p) {
  constexpr int NT = COUT / 16;                    // 16-channel row tiles
  constexpr int CHUNKS = COUT * 2 / 16;            // 16-byte chunks per output pixel
  constexpr int SROW = COUT * 2 + 16;              // padded slab row: 16 pixel rows fall on different banks
  constexpr int SLAB = 16 * SROW;                  // one wave's transpose slab
  __shared__ __attribute__((aligned(16))) char patch[ST_PATCH_BYTES];
  __shared__ __attribute__((aligned(16))) char slab[4][SLAB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, fq = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;

  // ---- weights -> A fragments (lane: out channel i*16 + col, K-step s: tap 4s + fq, 8 channels)
  bf16x8 wf[NT][3];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = 4 * s + fq;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (tap < 9) v = *reinterpret_cast<const u32x4*>(p.w + ((long long)(i * 16 + col) * 9 + tap) * 8);
      wf[i][s] = as_bf16x8(v);
    }
  f32x4 bv[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    bv[i] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + i * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- stage the input patch: rows 2*oy0-1 .. +16, columns 2*ox0-1 .. +128 (zero outside the image)
  const bf16_t* xb = p.x + (long long)b * p.H * p.W * 8;
  const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
  for (int idx = tid; idx < ST_PH * ST_PW; idx += 256) {
    const int r = idx / ST_PW, q = idx - r * ST_PW;
    const int iy = iy0 + r, ix = ix0 + q;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = *reinterpret_cast<const u32x4*>(xb + ((long long)iy * p.W + ix) * 8);
    const int slot = r * ST_PW + ((q & 1) ? ST_EVEN + (q >> 1) : (q >> 1));
    *reinterpret_cast<u32x4*>(&patch[slot * 16]) = v;
  }
  __syncthreads();

  // per-lane LDS offsets of the three K-steps (tap >= 9 meets zero weights: any valid address will do)
  int boff[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = min(4 * s + fq, 8);
    const int ky = tap / 3, kx = tap - 3 * ky;
    boff[s] = (ky * ST_PW + (kx == 1 ? ST_EVEN : (kx >> 1))) * 16;
  }

  char* my = slab[wave];
  // wave w owns output rows 2w, 2w+1 of the tile, 4 column groups of 16 pixels each
#pragma unroll 1
  for (int g = 0; g < 8; ++g) {
    const int ry = wave * 2 + (g >> 2), cg = (g & 3) * 16;
    const int oy = oy0 + ry;
    if (oy >= p.Ho || ox0 + cg >= p.Wo) continue;      // uniform per wave
    const int base = ((2 * ry) * ST_PW + cg + col) * 16;
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = bv[i];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&patch[base + boff[s]]);
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][s], bfr, acc[i], 0, 0, 0);
    }
    // activation, bf16, transpose through the wave's slab: pixel `col` row, channels i*16 + fq*4 .. +3
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (bf16_t)dlamd_act(acc[i][j], p.act);
      *reinterpret_cast<bf16x4*>(&my[col * SROW + (i * 16 + fq * 4) * 2]) = o;
    }
    bf16_t* yrow = p.y + ((long long)(b * p.Ho + oy) * p.Wo + ox0 + cg) * p.y_cstride + p.y_coff;
    const int npx = min(16, p.Wo - (ox0 + cg));
#pragma unroll
    for (int k = lane; k < 16 * CHUNKS; k += 64) {
      const int px = k / CHUNKS, ch = k - px * CHUNKS;
      const u32x4 v = *reinterpret_cast<const u32x4*>(&my[px * SROW + ch * 16]);
      if (px < npx) *reinterpret_cast<u32x4*>(yrow + (long long)px * p.y_cstride + ch * 8) = v;
    }
  }
}

}  // namespace

// Host contract (checked by the caller in conv_igemm.hip: dlamd_conv2d): Cin == x_cstride == 8, 3x3, stride 2, pad 1,
// plain bf16 output, no residual, Cout in {32, 64}, y_cstride % 8 == 0, y_coff % 8 == 0.
int dlamd_conv_stem_c8s2(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Ho, int Wo, int Cout,
                         int act, int y_cstride, int y_coff, hipStream_t stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Ho != (H - 1) / 2 + 1 || Wo != (W - 1) / 2 + 1) return -21;
  if ((Cout != 32 && Cout != 64) || y_cstride % 8 != 0 || y_coff % 8 != 0 || y_coff + Cout > y_cstride) return -22;
  StemArgs a;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y;
  a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.act = act; a.y_cstride = y_cstride; a.y_coff = y_coff;
  a.tiles_x = (Wo + ST_TW - 1) / ST_TW;
  a.tiles_y = (Ho + ST_TH - 1) / ST_TH;
  const long long tiles = (long long)B * a.tiles_x * a.tiles_y;
  if (tiles <= 0 || tiles > 0x7fffffffLL) return -23;
  dim3 grid((unsigned)tiles), block(256);
  if (Cout == 64) hipLaunchKernelGGL(conv_stem_c8s2_kernel<64>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(conv_stem_c8s2_kernel<32>, grid, block, 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -24;
}
