// Baseline JPEG decode on the GPU (SURVEY N2 "decode page image", hard part 7.4-7).
//
// A 300-DPI A4 scan is 8.7 Mpixel; libjpeg-turbo needs ~55 ms per page per CPU
// core, which caps a 16-core host at ~280 pages/s and would starve the GPU
// pipeline (~2-3 ms per page).  Huffman decoding is sequential only *within* a
// restart segment, so every restart segment gets its own thread here:
//
//   1. jpeg_huffman_kernel   one lane per restart segment: bit reader with
//                            byte-unstuffing, 10-bit LUT decode (tables in LDS),
//                            DC prediction, dequantisation, de-zigzag; writes the
//                            non-zero coefficients of the luma plane (int16).
//   2. jpeg_idct_kernel      separable float IDCT, 64 lanes = 8 blocks x 8
//                            rows/cols, transposes through LDS, packed 8-byte
//                            u8 row stores.
//
// Output is the luma plane (gray) - all the OCR models consume.  Chroma blocks of
// YCbCr files are entropy-decoded (to stay in sync) and discarded.
// Files without restart markers / progressive files are decoded on the host
// thread pool instead (ops/jpeg.py).
#include "common.h"

#include <stdlib.h>

#define JPEG_DEFAULT_SPREAD 1
#define JPEG_LUT_BITS 10
#define JPEG_LUT_SIZE (1 << JPEG_LUT_BITS)

struct JpegTables {
  uint16_t qt[3][64];
  uint16_t lut[4][JPEG_LUT_SIZE];
  int32_t maxcode[4][18];
  int32_t valoff[4][17];
  uint8_t vals[4][256];
};
static_assert(sizeof(JpegTables) == 3 * 64 * 2 + 4 * JPEG_LUT_SIZE * 2 + 4 * 18 * 4 + 4 * 17 * 4 + 4 * 256, "ABI");

struct JpegImageDesc {
  const uint8_t* data;        // JPEG stream (>= 16 readable bytes of slack after scan_end)
  const uint32_t* seg_off;    // [n_segments] offsets into data
  const JpegTables* tables;
  int16_t* coef;              // [Hb * Wb * 64] luma coefficients, zero-filled by the caller
  uint8_t* out;               // [Hb * 8, Wb * 8] gray
  int32_t n_segments, restart_interval, mcus_x, total_mcus;
  int32_t ncomp, hs0, vs0, scan_end;
  int32_t Wb, Hb, pad0, pad1;
};
static_assert(sizeof(JpegImageDesc) == 88, "ABI");

__constant__ uint8_t c_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc;   // MSB-aligned
  int nbits;
  bool hit_marker;

  __device__ __forceinline__ void init(const uint8_t* s, const uint8_t* e) {
    p = s; end = e; acc = 0; nbits = 0; hit_marker = false;
  }
  // guarantee >= 32 valid bits (zero-extended past the end of the segment)
  __device__ __forceinline__ void refill() {
    if (nbits > 32) return;
    if (!hit_marker && p + 4 <= end) {
      uint32_t w;
      __builtin_memcpy(&w, p, 4);
      uint32_t x = ~w;
      if ((((x - 0x01010101u) & ~x) & 0x80808080u) == 0u) {  // no 0xFF byte among the four
        uint32_t be = __builtin_bswap32(w);
        acc |= (uint64_t)be << (32 - nbits);
        nbits += 32;
        p += 4;
        return;
      }
    }
    while (nbits <= 56) {
      uint32_t b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          uint32_t b2 = (p + 1 < end) ? p[1] : 0xD9;
          if (b2 == 0) p += 2;
          else { hit_marker = true; b = 0; }
        } else {
          ++p;
        }
      }
      acc |= (uint64_t)b << (56 - nbits);
      nbits += 8;
    }
  }
  __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(acc >> (64 - n)); }
  __device__ __forceinline__ void skip(int n) { acc <<= n; nbits -= n; }
};

__device__ __forceinline__ int huff_decode(BitReader& br, const uint16_t* lut, const JpegTables* __restrict__ tb, int t) {
  uint32_t e = lut[br.peek(JPEG_LUT_BITS)];
  if (e) {
    br.skip(e >> 8);
    return e & 0xff;
  }
  // slow path: codes of 11..16 bits (canonical decode)
  int len = JPEG_LUT_BITS + 1;
  int code = (int)br.peek(len);
  while (len <= 16 && code > tb->maxcode[t][len]) {
    ++len;
    code = (int)br.peek(len);
  }
  if (len > 16) { br.skip(16); return 0; }
  br.skip(len);
  return tb->vals[t][(code + tb->valoff[t][len]) & 0xff];
}

__device__ __forceinline__ int receive_extend(BitReader& br, int s) {
  if (s == 0) return 0;
  int v = (int)br.peek(s);
  br.skip(s);
  return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

// `spread` = lanes per segment slot: only every spread-th lane decodes (1: all 64 lanes of a wave busy).  The decode
// loops are data dependent, so a wave runs the union of its lanes' paths; fewer segments per wave cut that
// divergence and spread the (few thousand) segments over more SIMDs.
__global__ __launch_bounds__(128) void jpeg_huffman_kernel(const JpegImageDesc* __restrict__ descs, int spread) {
  __shared__ uint16_t s_lut[4][JPEG_LUT_SIZE];
  __shared__ uint16_t s_qt[64];
  __shared__ uint8_t s_zz[64];
  const JpegImageDesc d = descs[blockIdx.y];
  const int per_block = 128 / spread;
  if ((int)(blockIdx.x * per_block) >= d.n_segments) return;
  const JpegTables* tb = d.tables;
  const int ntab = d.ncomp == 1 ? 2 : 4;
  for (int i = threadIdx.x; i < ntab * JPEG_LUT_SIZE; i += 128) (&s_lut[0][0])[i] = (&tb->lut[0][0])[i];
  if (threadIdx.x < 64) { s_qt[threadIdx.x] = tb->qt[0][threadIdx.x]; s_zz[threadIdx.x] = c_zigzag[threadIdx.x]; }
  __syncthreads();
  const int seg = blockIdx.x * per_block + threadIdx.x / spread;
  if (seg >= d.n_segments || (threadIdx.x % spread) != 0) return;
  const uint32_t start = d.seg_off[seg];
  // a segment ends 2 bytes before the next one starts (the RSTn marker) or at the end of the scan
  const uint32_t stop = (seg + 1 < d.n_segments) ? d.seg_off[seg + 1] - 2 : (uint32_t)d.scan_end;
  BitReader br;
  br.init(d.data + start, d.data + stop);
  const int ri = d.restart_interval > 0 ? d.restart_interval : d.total_mcus;
  const int mcu0 = seg * ri;
  const int mcu1 = min(mcu0 + ri, d.total_mcus);
  const int yblocks = d.hs0 * d.vs0;
  const int nblk = d.ncomp == 1 ? 1 : yblocks + 2;
  int pred[3] = {0, 0, 0};
  for (int m = mcu0; m < mcu1; ++m) {
    const int my = m / d.mcus_x, mx = m - my * d.mcus_x;
    for (int b = 0; b < nblk; ++b) {
      const bool luma = b < yblocks;
      const int comp = luma ? 0 : (b - yblocks + 1);
      const int tdc = luma ? 0 : 2, tac = luma ? 1 : 3;
      int16_t* cb = nullptr;
      if (luma) {
        int bx = mx * d.hs0 + (b % d.hs0), by = my * d.vs0 + (b / d.hs0);
        cb = d.coef + ((long long)by * d.Wb + bx) * 64;
      }
      br.refill();
      int s = huff_decode(br, s_lut[tdc], tb, tdc);
      br.refill();
      pred[comp] += receive_extend(br, s & 15);
      if (luma) {
        int v = pred[0] * (int)s_qt[0];
        cb[0] = (int16_t)max(-32768, min(32767, v));
      }
      for (int k = 1; k < 64;) {
        br.refill();
        int rs = huff_decode(br, s_lut[tac], tb, tac);
        int r = rs >> 4, sz = rs & 15;
        if (sz == 0) {
          if (r == 15) { k += 16; continue; }
          break;  // EOB
        }
        k += r;
        if (k > 63) break;
        int v = receive_extend(br, sz);
        if (luma) {
          int z = s_zz[k];
          int q = v * (int)s_qt[z];
          cb[z] = (int16_t)max(-32768, min(32767, q));
        }
        ++k;
      }
    }
  }
}

// 1-D 8-point inverse DCT (direct form): out[x] = 1/2 * sum_u c(u) in[u] cos((2x+1) u pi / 16)
__device__ __forceinline__ void idct8(const float* in, float* out) {
  const float C0 = 0.35355339059327379f;
  const float c1 = 0.49039264020161522f, c2 = 0.46193976625564337f, c3 = 0.41573480615127262f;
  const float c4 = 0.35355339059327379f, c5 = 0.27778511650980114f, c6 = 0.19134171618254492f;
  const float c7 = 0.09754516100806417f;
  // even part
  float e0 = C0 * in[0] + c4 * in[4], e1 = C0 * in[0] - c4 * in[4];
  float e2 = c2 * in[2] + c6 * in[6], e3 = c6 * in[2] - c2 * in[6];
  float a0 = e0 + e2, a3 = e0 - e2, a1 = e1 + e3, a2 = e1 - e3;
  // odd part
  float o0 = c1 * in[1] + c3 * in[3] + c5 * in[5] + c7 * in[7];
  float o1 = c3 * in[1] - c7 * in[3] - c1 * in[5] - c5 * in[7];
  float o2 = c5 * in[1] - c1 * in[3] + c7 * in[5] + c3 * in[7];
  float o3 = c7 * in[1] - c5 * in[3] + c3 * in[5] - c1 * in[7];
  out[0] = a0 + o0; out[7] = a0 - o0;
  out[1] = a1 + o1; out[6] = a1 - o1;
  out[2] = a2 + o2; out[5] = a2 - o2;
  out[3] = a3 + o3; out[4] = a3 - o3;
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const JpegImageDesc* __restrict__ descs) {
  // 256 threads = 32 DCT blocks x 8 rows.  LDS: floats [32][8][9] (padded rows)
  __shared__ float s_t[32][8][9];
  const JpegImageDesc d = descs[blockIdx.y];
  const long long nblocks = (long long)d.Wb * d.Hb;
  const int lb = threadIdx.x >> 3, r = threadIdx.x & 7;
  const long long blk = (long long)blockIdx.x * 32 + lb;
  const bool ok = blk < nblocks;
  float in[8], tmp[8];
  if (ok) {
    // one 16-byte load = row r of the block
    const u32x4 raw = *reinterpret_cast<const u32x4*>(d.coef + blk * 64 + r * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      in[2 * k] = (float)(int16_t)(raw[k] & 0xffffu);
      in[2 * k + 1] = (float)(int16_t)(raw[k] >> 16);
    }
    idct8(in, tmp);  // row pass (along u -> x)
#pragma unroll
    for (int k = 0; k < 8; ++k) s_t[lb][r][k] = tmp[k];
  }
  __syncthreads();
  if (ok) {
    // column pass: thread (lb, r) now owns column r
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = s_t[lb][k][r];
    idct8(in, tmp);
  }
  __syncthreads();
  if (ok) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s_t[lb][k][r] = tmp[k];
  }
  __syncthreads();
  if (ok) {
    // thread (lb, r) stores output row r of its block: 8 bytes
    uint32_t w0 = 0, w1 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = s_t[lb][r][k] + 128.f;
      int iv = (int)__builtin_rintf(v);
      iv = max(0, min(255, iv));
      if (k < 4) w0 |= (uint32_t)iv << (8 * k);
      else w1 |= (uint32_t)iv << (8 * (k - 4));
    }
    const int by = (int)(blk / d.Wb), bx = (int)(blk - (long long)by * d.Wb);
    uint8_t* o = d.out + ((long long)(by * 8 + r) * d.Wb + bx) * 8;
    *reinterpret_cast<u32x2*>(o) = u32x2{w0, w1};
  }
}

// descs: device array of n JpegImageDesc; max_segments / max_blocks: maxima over the batch (grid sizing).
extern "C" int dlamd_jpeg_decode(const void* descs, int n, int max_segments, long long max_blocks, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n > 65535 || max_segments <= 0 || max_blocks <= 0) return -1;
  static const int spread_env = getenv("DLAMD_JPEG_SPREAD") ? atoi(getenv("DLAMD_JPEG_SPREAD")) : 0;
  int spread = JPEG_DEFAULT_SPREAD;
  if (spread_env == 1 || spread_env == 2 || spread_env == 4 || spread_env == 8 || spread_env == 16 || spread_env == 32 ||
      spread_env == 64)
    spread = spread_env;
  const int per_block = 128 / spread;
  dim3 g1((max_segments + per_block - 1) / per_block, n);
  hipLaunchKernelGGL(jpeg_huffman_kernel, g1, dim3(128), 0, stream, (const JpegImageDesc*)descs, spread);
  long long gb = (max_blocks + 31) / 32;
  if (gb > 0x7fffffffLL) return -2;
  dim3 g2((unsigned)gb, n);
  hipLaunchKernelGGL(jpeg_idct_kernel, g2, dim3(256), 0, stream, (const JpegImageDesc*)descs);
  return DLAMD_CHECK_LAUNCH();
}
