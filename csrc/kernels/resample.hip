// Page raster -> tensor and line crop + resize.
//
// One fused kernel: for every output item (a whole page for detection, a text
// line box for recognition) read a rectangular region of a u8 source image
// (gray or RGB, HWC), resample it to the requested size (fractional box filter
// when shrinking - 300-DPI pages are shrunk ~2.6x for detection, plain bilinear
// would alias - bilinear when enlarging), normalise per channel, convert to
// bf16 and write NHWC with the channel dimension padded to 8 so that the first
// convolution sees 16-byte pixels.  Pixels outside the valid output size are
// written as zeros (bucket padding).  Optional 90-degree rotation for tall
// (vertical) boxes, and an arbitrary angle about the region's centre for the
// slanted text lines of a skewed scan (affine crop: up to 4 x 4 bilinear taps
// per output pixel stand in for the box filter).
//
// Each thread produces one output pixel = one 16-byte store; neighbouring lanes read neighbouring source spans.
// The common windows (box filter up to 4 x 4 taps, affine crop up to 2 x 2 samples) are unrolled with zero
// weights for the taps a pixel does not use: a data-dependent tap loop serialises one memory latency per tap and
// ran the page resample at a quarter of the HBM rate.
#include "common.h"

struct ResampleItem {
  const uint8_t* src;   // HWC u8
  int Hs, Ws, C;        // source dims, C in {1, 3}
  int rot;              // 0 none, 1 = rotate region 90 deg ccw into output
  float x0, y0, x1, y1; // source region (pixels, half-open, may be fractional)
  int out_h, out_w;     // valid output size (<= Hd, Wd)
  int pitch;            // source row pitch in bytes (0: Ws * C)
  float angle;          // radians; != 0: the region is the rectangle x0..y1 turned by this angle about its centre
};
static_assert(sizeof(ResampleItem) == 56, "ResampleItem layout is part of the Python ABI");

__device__ __forceinline__ void box_axis(float lo, float hi, int n, int& i0, int& i1) {
  i0 = max(0, (int)floorf(lo));
  i1 = min(n, (int)ceilf(hi));
  if (i1 <= i0) { i0 = min(max(i0, 0), n - 1); i1 = i0 + 1; }
}

// Mean source value (0..255) of output pixel (ox, oy) of item `it`, per channel (a gray source fills all three).
// The caller has checked ox < out_w, oy < out_h and src != nullptr.
__device__ __forceinline__ void resample_pixel(const ResampleItem& it, const int ox, const int oy, float (&acc)[3]) {
  // map output pixel -> source span
  float rw = it.x1 - it.x0, rh = it.y1 - it.y0;
  float u, v, su, sv;  // output coords along source x / y and scales
  if (it.rot == 0) {
    su = rw / it.out_w; sv = rh / it.out_h;
    u = ox; v = oy;
  } else {
    // output x runs along source -y (ccw rotation): out_w spans region height
    su = rw / it.out_h; sv = rh / it.out_w;
    u = oy; v = (it.out_w - 1 - ox);
  }
  acc[0] = 0.f; acc[1] = 0.f; acc[2] = 0.f;
  const int C = it.C;
  const uint8_t* src = it.src;
  const long long pitch = it.pitch > 0 ? it.pitch : (long long)it.Ws * C;
  if (it.angle != 0.f) {
    // affine crop: sample positions are laid out in the rectangle's own frame and turned into the source frame
    const float ca = cosf(it.angle), sa = sinf(it.angle);
    const float cx = 0.5f * (it.x0 + it.x1), cy = 0.5f * (it.y0 + it.y1);
    const int nx = min(4, max(1, (int)ceilf(su))), ny = min(4, max(1, (int)ceilf(sv)));
    if (nx <= 2 && ny <= 2 && C == 1) {
      // shrink factors up to 2 (text lines of ordinary sizes): the 2 x 2 sample grid is unrolled so that its 16 loads
      // are in flight together; samples beyond nx / ny get weight 0 and read the first sample's address
      float wgt[4];
      const uint8_t* pa[4][4];
      float fxs[4], fys[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = t & 1, j = t >> 1;
        const bool on = i < nx && j < ny;
        const int ii = on ? i : 0, jj = on ? j : 0;
        const float ly = (v + (jj + 0.5f) / ny) * sv - 0.5f * rh;
        const float lx = (u + (ii + 0.5f) / nx) * su - 0.5f * rw;
        float px = cx + lx * ca - ly * sa - 0.5f, py = cy + lx * sa + ly * ca - 0.5f;
        px = fminf(fmaxf(px, 0.f), (float)(it.Ws - 1));
        py = fminf(fmaxf(py, 0.f), (float)(it.Hs - 1));
        const int xa = (int)floorf(px), ya = (int)floorf(py);
        const int xb = min(xa + 1, it.Ws - 1), yb = min(ya + 1, it.Hs - 1);
        fxs[t] = px - xa; fys[t] = py - ya;
        const uint8_t* r0 = src + (long long)ya * pitch;
        const uint8_t* r1 = src + (long long)yb * pitch;
        pa[t][0] = r0 + xa; pa[t][1] = r0 + xb; pa[t][2] = r1 + xa; pa[t][3] = r1 + xb;
        wgt[t] = on ? 1.f : 0.f;
      }
      float pv[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[t][q] = (float)*pa[t][q];
      // same order of additions as the general loop (row by row over the sample grid)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float fx = fxs[t], fy = fys[t];
        const float val = (pv[t][0] * (1.f - fx) + pv[t][1] * fx) * (1.f - fy) + (pv[t][2] * (1.f - fx) + pv[t][3] * fx) * fy;
        if (wgt[t] != 0.f) acc[0] += val;
      }
    } else {
      for (int j = 0; j < ny; ++j) {
        const float ly = (v + (j + 0.5f) / ny) * sv - 0.5f * rh;
        for (int i = 0; i < nx; ++i) {
          const float lx = (u + (i + 0.5f) / nx) * su - 0.5f * rw;
          float px = cx + lx * ca - ly * sa - 0.5f, py = cy + lx * sa + ly * ca - 0.5f;
          px = fminf(fmaxf(px, 0.f), (float)(it.Ws - 1));
          py = fminf(fmaxf(py, 0.f), (float)(it.Hs - 1));
          const int xa = (int)floorf(px), ya = (int)floorf(py);
          const int xb = min(xa + 1, it.Ws - 1), yb = min(ya + 1, it.Hs - 1);
          const float fx = px - xa, fy = py - ya;
          const uint8_t* r0 = src + (long long)ya * pitch;
          const uint8_t* r1 = src + (long long)yb * pitch;
          for (int ch = 0; ch < C; ++ch) {
            const float a = r0[xa * C + ch], b2 = r0[xb * C + ch], c2 = r1[xa * C + ch], d = r1[xb * C + ch];
            acc[ch] += (a * (1.f - fx) + b2 * fx) * (1.f - fy) + (c2 * (1.f - fx) + d * fx) * fy;
          }
        }
      }
    }
    const float inv = 1.f / (float)(nx * ny);
    acc[0] *= inv; acc[1] *= inv; acc[2] *= inv;
  } else if (su > 1.f || sv > 1.f) {
    // fractional box filter over [lo, hi)
    float xlo = it.x0 + u * su, xhi = xlo + su;
    float ylo = it.y0 + v * sv, yhi = ylo + sv;
    int xa, xb, ya, yb;
    box_axis(xlo, xhi, it.Ws, xa, xb);
    box_axis(ylo, yhi, it.Hs, ya, yb);
    float wsum = 0.f;
    if (C == 1 && xb - xa <= 4 && yb - ya <= 4) {
      // shrink factors up to 3 (300-DPI pages to the detector's size, text lines to 48 rows): a fixed 4 x 4 window,
      // unrolled, so that the 16 byte loads are issued back to back instead of one per trip of a data-dependent
      // loop (the loop form waits out a memory latency per tap); taps outside [xa, xb) x [ya, yb) weigh 0 and
      // read a clamped address
      float wx[4], wy[4];
      int xo[4];
      const uint8_t* rows[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = xa + i, y = ya + i;
        wx[i] = x < xb ? fmaxf(fminf(xhi, (float)(x + 1)) - fmaxf(xlo, (float)x), 1e-6f) : 0.f;
        wy[i] = y < yb ? fmaxf(fminf(yhi, (float)(y + 1)) - fmaxf(ylo, (float)y), 1e-6f) : 0.f;
        xo[i] = min(x, xb - 1);
        rows[i] = src + (long long)min(y, yb - 1) * pitch;
      }
      float pv[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) pv[j][i] = (float)rows[j][xo[i]];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float w = wx[i] * wy[j];
          wsum += w;
          acc[0] += w * pv[j][i];
        }
    } else {
      for (int y = ya; y < yb; ++y) {
        float wy = fminf(yhi, (float)(y + 1)) - fmaxf(ylo, (float)y);
        wy = fmaxf(wy, 1e-6f);
        const uint8_t* row = src + (long long)y * pitch;
        for (int x = xa; x < xb; ++x) {
          float wx = fminf(xhi, (float)(x + 1)) - fmaxf(xlo, (float)x);
          wx = fmaxf(wx, 1e-6f);
          float w = wx * wy;
          wsum += w;
          if (C == 1) {
            acc[0] += w * row[x];
          } else {
            acc[0] += w * row[x * 3];
            acc[1] += w * row[x * 3 + 1];
            acc[2] += w * row[x * 3 + 2];
          }
        }
      }
    }
    float inv = 1.f / wsum;
    acc[0] *= inv; acc[1] *= inv; acc[2] *= inv;
  } else {
    // bilinear, pixel centres (align_corners = False)
    float sx = it.x0 + (u + 0.5f) * su - 0.5f;
    float sy = it.y0 + (v + 0.5f) * sv - 0.5f;
    sx = fminf(fmaxf(sx, 0.f), (float)(it.Ws - 1));
    sy = fminf(fmaxf(sy, 0.f), (float)(it.Hs - 1));
    int xa = (int)floorf(sx), ya = (int)floorf(sy);
    int xb = min(xa + 1, it.Ws - 1), yb = min(ya + 1, it.Hs - 1);
    float fx = sx - xa, fy = sy - ya;
    const uint8_t* r0 = src + (long long)ya * pitch;
    const uint8_t* r1 = src + (long long)yb * pitch;
    for (int ch = 0; ch < C; ++ch) {
      float a = r0[xa * C + ch], b = r0[xb * C + ch], c2 = r1[xa * C + ch], d = r1[xb * C + ch];
      acc[ch] = (a * (1.f - fx) + b * fx) * (1.f - fy) + (c2 * (1.f - fx) + d * fx) * fy;
    }
  }
  if (C == 1) { acc[1] = acc[0]; acc[2] = acc[0]; }
}

__global__ __launch_bounds__(256) void resample_kernel(const ResampleItem* __restrict__ items, bf16_t* __restrict__ out,
                                                       int Hd, int Wd, float m0, float m1, float m2, float s0,
                                                       float s1, float s2) {
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z;
  if (ox >= Wd || oy >= Hd) return;
  const ResampleItem it = items[n];
  bf16_t* dst = out + (((long long)n * Hd + oy) * Wd + ox) * 8;
  u32x4 zero = {0u, 0u, 0u, 0u};
  if (ox >= it.out_w || oy >= it.out_h || it.src == nullptr) {
    *reinterpret_cast<u32x4*>(dst) = zero;
    return;
  }
  float acc[3];
  resample_pixel(it, ox, oy, acc);
  const float k = 1.f / 255.f;
  bf16x8 o;
  o[0] = (bf16_t)((acc[0] * k - m0) * s0);
  o[1] = (bf16_t)((acc[1] * k - m1) * s1);
  o[2] = (bf16_t)((acc[2] * k - m2) * s2);
  o[3] = (bf16_t)0.f; o[4] = (bf16_t)0.f; o[5] = (bf16_t)0.f; o[6] = (bf16_t)0.f; o[7] = (bf16_t)0.f;
  *reinterpret_cast<bf16x8*>(dst) = o;
}

// items: device array of N ResampleItem; out: [N, Hd, Wd, 8] bf16.
// mean/inv_std are per channel: out = (v/255 - mean) * inv_std.
extern "C" int dlamd_resample(const void* items, int N, void* out, int Hd, int Wd, float m0, float m1, float m2,
                              float s0, float s1, float s2, hipStream_t stream) {
  if (N <= 0) return 0;
  if (N > 65535 || Hd <= 0 || Wd <= 0) return -1;
  dim3 grid((Wd + 63) / 64, (Hd + 3) / 4, N);
  if (grid.y > 65535) return -2;
  hipLaunchKernelGGL(resample_kernel, grid, dim3(256), 0, stream, (const ResampleItem*)items, (bf16_t*)out, Hd, Wd,
                     m0, m1, m2, s0, s1, s2);
  return DLAMD_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------
// Resample fused with the first convolution (3 x 3, stride 2, pad 1, 3 -> COUT channels, bias + activation): the
// [N, Hd, Wd, 8] tensor of the plain kernel - 16 bytes per pixel, 10 of them channel padding - is never written or
// read; the detector's first layer on a batch of 25 pages skips 2 x 516 MB that way, the recogniser's 2 x 16 bytes per
// crop pixel.
//
// A workgroup owns 8 x 32 output pixels.  Phase 1: the (2*8+1) x (2*32+1) resampled, normalised, bf16-rounded pixels
// under that tile go to LDS (zeros outside the item's valid size, which is also the convolution's padding).  Phase 2:
// every wave takes two output rows; a 16-pixel group is one MFMA column block whose K index is tap * 3 + channel
// (27 of 32 used), gathered from the LDS tile, against the COUT x 27 weights held in registers.  Results go through a
// per-wave LDS slab so that the 16 pixels x COUT channels leave as one contiguous run of 16-byte stores.
struct StemFuseArgs {
  const ResampleItem* items;
  const bf16_t* w;     // [COUT][3][3][8] bf16 (input channels padded to 8, BN folded)
  const float* bias;   // [COUT]
  bf16_t* y;           // [N, Ho, Wo, COUT]
  int Hd, Wd, Ho, Wo, act;
  float m0, m1, m2, s0, s1, s2;
};

static constexpr int SF_TH = 8, SF_TW = 32;
static constexpr int SF_RH = 2 * SF_TH + 1, SF_RW = 2 * SF_TW + 1;

template <int COUT>
__global__ __launch_bounds__(256) void resample_stem_kernel(const StemFuseArgs p) {
  constexpr int CT = COUT / 16;                 // 16-channel tiles
  constexpr int SLAB_ROW = COUT * 2 + 16;       // bytes per pixel in the store slab
  __shared__ __align__(16) u32x2 tile[SF_RH * SF_RW];
  __shared__ __align__(16) char slabs[4 * 16 * SLAB_ROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.z;
  const int oy0 = blockIdx.y * SF_TH, ox0 = blockIdx.x * SF_TW;
  const ResampleItem it = p.items[n];

  // phase 1
  const float k255 = 1.f / 255.f;
  for (int e = tid; e < SF_RH * SF_RW; e += 256) {
    const int ty = e / SF_RW, tx = e - ty * SF_RW;
    const int ry = 2 * oy0 - 1 + ty, rx = 2 * ox0 - 1 + tx;
    u32x2 v = {0u, 0u};
    if (ry >= 0 && rx >= 0 && ry < it.out_h && rx < it.out_w && ry < p.Hd && rx < p.Wd && it.src != nullptr) {
      float acc[3];
      resample_pixel(it, rx, ry, acc);
      const bf16x4 o = {(bf16_t)((acc[0] * k255 - p.m0) * p.s0), (bf16_t)((acc[1] * k255 - p.m1) * p.s1),
                        (bf16_t)((acc[2] * k255 - p.m2) * p.s2), (bf16_t)0.f};
      v = __builtin_bit_cast(u32x2, o);
    }
    tile[e] = v;
  }

  // weights as MFMA A fragments: row = channel (lane & 15) of tile ct, k = 8 * (lane >> 4) + j -> (tap, c) = (k / 3, k % 3)
  const int col = lane & 15, kq = lane >> 4;
  bf16x8 wf[CT];
  int koff[8];          // byte offset of element j's source pixel / channel relative to the output pixel's window origin
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * kq + j;
    const int tap = k / 3, c = k - tap * 3;
    const int dy = tap / 3, dx = tap - dy * 3;
    koff[j] = k < 27 ? ((dy * SF_RW + dx) * 8 + c * 2) : -1;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      wf[ct][j] = k < 27 ? p.w[((size_t)(ct * 16 + col) * 9 + tap) * 8 + c] : (bf16_t)0.f;
  }
  f32x4 bias4[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) bias4[ct][q] = p.bias ? p.bias[ct * 16 + kq * 4 + q] : 0.f;
  __syncthreads();

  // phase 2
  char* slab = slabs + wave * 16 * SLAB_ROW;
  const char* tbytes = reinterpret_cast<const char*>(tile);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int ly = 2 * wave + (g >> 1), lx0 = (g & 1) * 16;
    const int oy = oy0 + ly;
    // window origin of this lane's pixel inside the tile: resampled (2 * ly, 2 * lx)
    const int org = ((2 * ly) * SF_RW + 2 * (lx0 + col)) * 8;
    bf16x8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bf16_t v = (bf16_t)0.f;
      if (koff[j] >= 0) v = *reinterpret_cast<const bf16_t*>(tbytes + org + koff[j]);
      b[j] = v;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], b, bias4[ct], 0, 0, 0);
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = (bf16_t)dlamd_act(acc[q], p.act);
      *reinterpret_cast<bf16x4*>(slab + col * SLAB_ROW + (ct * 16 + kq * 4) * 2) = o;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // 16 pixels x COUT channels = 16 * COUT * 2 bytes, contiguous in y when the pixels are; 16 bytes per lane and trip
    constexpr int CHUNKS_PER_PX = COUT * 2 / 16;
#pragma unroll
    for (int i = 0; i < 16 * CHUNKS_PER_PX / 64; ++i) {
      const int c = i * 64 + lane, px = c / CHUNKS_PER_PX, cc = c - px * CHUNKS_PER_PX;
      const u32x4 v = *reinterpret_cast<const u32x4*>(slab + px * SLAB_ROW + cc * 16);
      const int ox = ox0 + lx0 + px;
      if (oy < p.Ho && ox < p.Wo)
        *reinterpret_cast<u32x4*>(p.y + (((size_t)n * p.Ho + oy) * p.Wo + ox) * COUT + cc * 8) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// items: N ResampleItem (device); w: [Cout][3][3][8] bf16; y: [N, Ho, Wo, Cout] bf16 with Ho = (Hd - 1) / 2 + 1,
// Wo = (Wd - 1) / 2 + 1; Cout 32 or 64.
extern "C" int dlamd_resample_stem(const void* items, int N, int Hd, int Wd, float m0, float m1, float m2, float s0,
                                   float s1, float s2, const void* w, const float* bias, void* y, int Cout, int act,
                                   hipStream_t stream) {
  if (N <= 0) return 0;
  if (N > 65535 || Hd <= 0 || Wd <= 0) return -1;
  if (Cout != 32 && Cout != 64) return -3;
  StemFuseArgs a;
  a.items = (const ResampleItem*)items;
  a.w = (const bf16_t*)w;
  a.bias = bias;
  a.y = (bf16_t*)y;
  a.Hd = Hd; a.Wd = Wd;
  a.Ho = (Hd - 1) / 2 + 1; a.Wo = (Wd - 1) / 2 + 1;
  a.act = act;
  a.m0 = m0; a.m1 = m1; a.m2 = m2; a.s0 = s0; a.s1 = s1; a.s2 = s2;
  dim3 grid((a.Wo + SF_TW - 1) / SF_TW, (a.Ho + SF_TH - 1) / SF_TH, N);
  if (grid.y > 65535) return -2;
  if (Cout == 32) hipLaunchKernelGGL(resample_stem_kernel<32>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(resample_stem_kernel<64>, grid, dim3(256), 0, stream, a);
  return DLAMD_CHECK_LAUNCH();
}
