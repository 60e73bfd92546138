// First-layer convolutions: 3x3, pad 1, small Cin - (Cin 8, stride 2) with RGB padded to 8 channels = one 16-byte pixel,
// and the Cin 32 layers that follow it in the detector stem (stride 1 and 2).
//
// The generic implicit-GEMM kernel gathers an im2col tile per K-tile; with K = 72 that is 9 x 16-byte gathers per
// output pixel for 2 MFMA K-steps of work, and the CRNN's first conv (8 M output pixels per 25-page batch) ran at
// 9 - 36 TFLOP/s, 20 % of all convolution time of the OCR bench (bench/conv_layer_breakdown.py, profiles/17).
// Here the layer is treated as what it is, a memory-bound stencil:
//   * a workgroup stages the input patch of an output tile (8 x 64 pixels; 17 x 129 x 16-byte patch for the Cin-8
//     layer) in LDS once, even and odd
//     input columns in separate planes so that the 16 pixels of an MFMA column group are contiguous for every tap;
//   * with K ordered (tap, channel) one MFMA B fragment row-group IS one pixel: lane (col, fq) of K-step s reads the
//     16-byte pixel of tap 4s+fq straight from the patch (ds_read_b128, conflict free) - no im2col copy at all;
//   * the whole weight tensor (Cout x 72, zero-padded to K = 96) lives in registers as A fragments;
//   * each wave transposes its 16-pixel x Cout results through a private LDS slab and stores whole 16-byte chunks.
// Same operand orientation as conv_igemm.hip (weights = A, pixels = B: a lane ends up with 4 consecutive channels).
#include "common.h"

namespace {


struct StemArgs {
  const bf16_t* x;      // [B, H, W, CIN]
  const bf16_t* w;      // [Cout, 3, 3, CIN]
  const float* bias;    // [Cout] or nullptr
  bf16_t* y;            // [B, Ho, Wo, y_cstride] written at channel offset y_coff
  int B, H, W, Ho, Wo;
  int act, y_cstride, y_coff;
  int tiles_x, tiles_y;
};

// CIN = 8 : K-step s covers taps 4s .. 4s+3 (lane group fq = tap 4s+fq, all 8 channels = the whole 16-byte pixel)
// CIN = 32: K-step s is tap s (lane group fq = channels 8fq .. 8fq+7 of that pixel)
template <int CIN, int S, int COUT, int TH, int TW>
__global__ __launch_bounds__(256) void conv_stem_kernel(StemArgs p) {
  constexpr int NT = COUT / 16;                    // 16-channel row tiles
  constexpr int KS = CIN == 8 ? 3 : 9;             // MFMA K-steps
  constexpr int PIXB = CIN * 2;                    // bytes per input pixel
  constexpr int PCH = PIXB / 16;                   // 16-byte chunks per input pixel
  constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
  constexpr int EVEN = TW + 1;                     // S == 2: even input columns first, then the odd ones
  constexpr int CHUNKS = COUT * 2 / 16;            // 16-byte chunks per output pixel
  constexpr int SROW = COUT * 2 + 16;              // padded slab row: 16 pixel rows fall on different banks
  constexpr int SLAB = 16 * SROW;                  // one wave's transpose slab
  constexpr int GPR = TW / 16;                     // MFMA column groups (16 pixels) per tile row
  constexpr int GPW = TH * GPR / 4;                // groups per wave
  static_assert(TW % 16 == 0 && (TH * GPR) % 4 == 0 && (CIN == 8 || CIN == 32) && (S == 1 || S == 2), "unsupported stem shape");
  __shared__ __attribute__((aligned(16))) char patch[PH * PW * PIXB];
  __shared__ __attribute__((aligned(16))) char slab[4][SLAB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, fq = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x; t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;

  // ---- weights -> A fragments (lane: out channel i*16 + col)
  bf16x8 wf[NT][KS];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if constexpr (CIN == 8) {
        const int tap = 4 * s + fq;
        if (tap < 9) v = *reinterpret_cast<const u32x4*>(p.w + ((long long)(i * 16 + col) * 9 + tap) * 8);
      } else {
        v = *reinterpret_cast<const u32x4*>(p.w + ((long long)(i * 16 + col) * 9 + s) * CIN + fq * 8);
      }
      wf[i][s] = as_bf16x8(v);
    }
  f32x4 bv[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
    bv[i] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + i * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- stage the input patch (zero outside the image)
  const bf16_t* xb = p.x + (long long)b * p.H * p.W * CIN;
  const int iy0 = S * oy0 - 1, ix0 = S * ox0 - 1;
  for (int idx = tid; idx < PH * PW * PCH; idx += 256) {
    const int pix = idx / PCH, sub = idx - pix * PCH;
    const int r = pix / PW, q = pix - r * PW;
    const int iy = iy0 + r, ix = ix0 + q;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
      v = *reinterpret_cast<const u32x4*>(xb + ((long long)iy * p.W + ix) * CIN + sub * 8);
    const int slot = S == 2 ? r * PW + ((q & 1) ? EVEN + (q >> 1) : (q >> 1)) : r * PW + q;
    *reinterpret_cast<u32x4*>(&patch[slot * PIXB + sub * 16]) = v;
  }
  __syncthreads();

  // per-lane LDS byte offsets of the K-steps, relative to (patch row S*ry, column group start + col)
  int boff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int tap = CIN == 8 ? min(4 * s + fq, 8) : s;     // CIN 8: taps >= 9 meet zero weights, any valid address will do
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int pc = S == 2 ? (kx == 1 ? EVEN : (kx >> 1)) : kx;
    boff[s] = (ky * PW + pc) * PIXB + (CIN == 8 ? 0 : fq * 16);
  }

  char* my = slab[wave];
#pragma unroll 1
  for (int g = 0; g < GPW; ++g) {
    const int gg = wave * GPW + g;
    const int ry = gg / GPR, cg = (gg % GPR) * 16;
    const int oy = oy0 + ry;
    if (oy >= p.Ho || ox0 + cg >= p.Wo) continue;      // uniform per wave
    const int base = ((S * ry) * PW + cg + col) * PIXB;
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = bv[i];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
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

template <int CIN, int S, int COUT, int TH, int TW>
int stem_launch(StemArgs a, hipStream_t stream) {
  a.tiles_x = (a.Wo + TW - 1) / TW;
  a.tiles_y = (a.Ho + TH - 1) / TH;
  const long long tiles = (long long)a.B * a.tiles_x * a.tiles_y;
  if (tiles <= 0 || tiles > 0x7fffffffLL) return -23;
  hipLaunchKernelGGL((conv_stem_kernel<CIN, S, COUT, TH, TW>), dim3((unsigned)tiles), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -24;
}

}  // namespace

// Served shapes (Cin, stride, Cout): (8, 2, 32 | 64), (32, 1, 32) - isolated 0.25 vs 0.44 ms, 0.27 vs 0.52 ms and
// 0.41 vs 0.69 ms against the best generic tile (bench/stem_probe.py).  A (32, 2, 64) instance measured 0.42 vs 0.34 ms
// (the 64-byte pixels leave too little LDS for occupancy) and was dropped.  Host contract (the caller in
// conv_igemm.hip: dlamd_conv2d checks the rest): x_cstride == Cin, 3x3, pad 1, plain bf16 output, no residual,
// y_cstride % 8 == 0, y_coff % 8 == 0.
bool dlamd_conv_stem_serves(int Cin, int stride, int Cout) {
  return (Cin == 8 && stride == 2 && (Cout == 32 || Cout == 64)) || (Cin == 32 && stride == 1 && Cout == 32);
}

int dlamd_conv_stem(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int stride, int Ho,
                    int Wo, int Cout, int act, int y_cstride, int y_coff, hipStream_t stream) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || !dlamd_conv_stem_serves(Cin, stride, Cout)) return -21;
  if (Ho != (H - 1) / stride + 1 || Wo != (W - 1) / stride + 1) return -21;
  if (y_cstride % 8 != 0 || y_coff % 8 != 0 || y_coff + Cout > y_cstride) return -22;
  StemArgs a;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y;
  a.B = B; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo;
  a.act = act; a.y_cstride = y_cstride; a.y_coff = y_coff;
  a.tiles_x = a.tiles_y = 0;
  if (Cin == 8) return Cout == 64 ? stem_launch<8, 2, 64, 8, 64>(a, stream) : stem_launch<8, 2, 32, 8, 64>(a, stream);
  return stem_launch<32, 1, 32, 8, 64>(a, stream);
}
