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
// Memory-bound: each thread produces one output pixel = one 16-byte store;
// neighbouring lanes read neighbouring source spans.
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
  float acc[3] = {0.f, 0.f, 0.f};
  const int C = it.C;
  const uint8_t* src = it.src;
  const long long pitch = it.pitch > 0 ? it.pitch : (long long)it.Ws * C;
  if (it.angle != 0.f) {
    // affine crop: sample positions are laid out in the rectangle's own frame and turned into the source frame
    const float ca = cosf(it.angle), sa = sinf(it.angle);
    const float cx = 0.5f * (it.x0 + it.x1), cy = 0.5f * (it.y0 + it.y1);
    const int nx = min(4, max(1, (int)ceilf(su))), ny = min(4, max(1, (int)ceilf(sv)));
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
