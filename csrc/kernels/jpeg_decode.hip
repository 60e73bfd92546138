// Baseline JPEG decode on the GPU.
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
//
// Streams WITHOUT restart markers (what most scanners write) have no entry points, so they are decoded by
// self-synchronisation (the property that a Huffman decoder started at a wrong bit falls into step with the true
// code-word sequence after a few symbols):
//   a. jpeg_sync_kernel      the scan is cut into 128-byte subsequences, one lane each.  Every lane decodes its
//                            subsequence from a guessed state (code-word boundary, start of a block), then keeps
//                            decoding into the following subsequences until its state at a subsequence end equals
//                            the one already recorded there - from then on both paths are identical.  Lower lanes
//                            overwrite higher ones, so within a 256-subsequence chunk everything follows the chunk's
//                            first lane, which has itself fallen into step long before the chunk ends.
//   b. jpeg_sync_seam_kernel one lane per chunk seam carries the end state of the previous chunk into the next one
//                            until it meets the recorded path, in passes that repeat until nothing changes; lane 0
//                            of the image starts from the real beginning, so by induction every recorded state is
//                            then the true one.
//   c. exclusive prefix sum of the blocks completed per subsequence (torch.cumsum) = where each lane writes.
//   d. jpeg_sync_decode_kernel  every lane decodes its subsequence once more from its true start state and writes
//                            dequantised AC coefficients and DC DIFFERENCES.
//   e. prefix sum of the DC differences over the luma blocks in decode order, jpeg_dc_apply_kernel, then the IDCT.
// Every bit is decoded about three times; in exchange a page yields ~10 000 independent lanes instead of one.
// Progressive files and exotic sampling are decoded on the host thread pool (ops/jpeg.py).
#include "common.h"

#include <stdlib.h>

#define JPEG_DEFAULT_SPREAD 1
#include "jpeg_core.h"

__constant__ uint8_t c_zigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

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

#define JPEG_SYNC_LOAD_TABLES(NT)                                                                          \
  __shared__ uint16_t s_lut[4][JPEG_LUT_SIZE];                                                             \
  __shared__ uint16_t s_qt[64];                                                                            \
  __shared__ uint8_t s_zz[64];                                                                             \
  {                                                                                                        \
    const int ntab = d.ncomp == 1 ? 2 : 4;                                                                 \
    for (int i = threadIdx.x; i < ntab * JPEG_LUT_SIZE; i += NT) (&s_lut[0][0])[i] = (&d.tables->lut[0][0])[i]; \
    if (threadIdx.x < 64) { s_qt[threadIdx.x] = d.tables->qt[0][threadIdx.x]; s_zz[threadIdx.x] = c_zigzag[threadIdx.x]; } \
  }                                                                                                        \
  __syncthreads();

__global__ __launch_bounds__(JPEG_SUB_CHUNK) void jpeg_sync_kernel(const JpegImageDesc* __restrict__ descs) {
  const JpegImageDesc d = descs[blockIdx.y];
  const int j0 = blockIdx.x * JPEG_SUB_CHUNK;
  if (j0 >= d.n_sub) return;
  JPEG_SYNC_LOAD_TABLES(JPEG_SUB_CHUNK)
  __shared__ uint32_t s_state[JPEG_SUB_CHUNK];
  __shared__ int s_cnt[JPEG_SUB_CHUNK];
  __shared__ int s_stuff[JPEG_SUB_CHUNK];
  const int tid = threadIdx.x, j = j0 + tid;
  const int n_here = min(JPEG_SUB_CHUNK, d.n_sub - j0);
  const bool valid = tid < n_here;
  s_stuff[tid] = valid ? sub_stuffing(d, j) : 0;
  __syncthreads();
  SubState st = {0, 0, 0};
  if (valid) {
    s_cnt[tid] = sub_decode<false>(d, j, st, s_stuff[tid], 0, s_lut, s_qt, s_zz);
    s_state[tid] = sub_pack(st);
  }
  __syncthreads();
  bool active = valid;
  for (int r = 1; r < JPEG_SUB_CHUNK; ++r) {
    const int t = tid + r;
    const bool go = active && t < n_here;
    uint32_t mine = 0;
    int cnt = 0;
    if (go) {
      cnt = sub_decode<false>(d, j + r, st, s_stuff[t], 0, s_lut, s_qt, s_zz);
      mine = sub_pack(st);
    }
    const bool same = go && s_state[t] == mine;
    __syncthreads();                                   // every comparison of this round is done before any write
    if (go) { s_state[t] = mine; s_cnt[t] = cnt; }     // the lower lane's path is the better one, merged or not
    active = go && !same;
    if (!__syncthreads_or(active ? 1 : 0)) break;
  }
  if (valid) {
    d.sub_state[j] = s_state[tid];
    d.sub_count[j] = s_cnt[tid];
  }
}

// Seams between chunks, one workgroup per image, one lane per seam (chunk c >= 1): walk into chunk c from the end
// state of chunk c-1 until the recorded path is met.  A pass in which some lane rewrote states may have invalidated
// the start of the following seam (only when a path crosses a whole chunk without merging: blank pages, whose
// identical code words never resynchronise), so passes repeat until one changes nothing - at most once per chunk,
// in which case the lanes have together done one serial decode of the scan.
__global__ __launch_bounds__(256) void jpeg_sync_seam_kernel(const JpegImageDesc* __restrict__ descs) {
  const JpegImageDesc d = descs[blockIdx.x];
  const int n_chunks = (d.n_sub + JPEG_SUB_CHUNK - 1) / JPEG_SUB_CHUNK;
  if (n_chunks <= 1) return;
  JPEG_SYNC_LOAD_TABLES(256)
  for (int pass = 0; pass < n_chunks; ++pass) {
    int changed = 0;
    for (int c = 1 + (int)threadIdx.x; c < n_chunks; c += 256) {
      const int j0 = c * JPEG_SUB_CHUNK;
      SubState st = sub_unpack(__hip_atomic_load(d.sub_state + j0 - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      const int n_here = min(JPEG_SUB_CHUNK, d.n_sub - j0);
      for (int t = 0; t < n_here; ++t) {
        const int cnt = sub_decode<false>(d, j0 + t, st, sub_stuffing(d, j0 + t), 0, s_lut, s_qt, s_zz);
        const uint32_t mine = sub_pack(st);
        if (d.sub_state[j0 + t] == mine) { d.sub_count[j0 + t] = cnt; break; }
        __hip_atomic_store(d.sub_state + j0 + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        d.sub_count[j0 + t] = cnt;
        changed = 1;
      }
    }
    __threadfence();
    if (!__syncthreads_or(changed)) break;
  }
}

__global__ __launch_bounds__(128) void jpeg_sync_decode_kernel(const JpegImageDesc* __restrict__ descs) {
  const JpegImageDesc d = descs[blockIdx.y];
  if ((int)(blockIdx.x * 128) >= d.n_sub) return;
  JPEG_SYNC_LOAD_TABLES(128)
  const int j = blockIdx.x * 128 + threadIdx.x;
  if (j >= d.n_sub) return;
  SubState st = j == 0 ? SubState{0, 0, 0} : sub_unpack(d.sub_state[j - 1]);
  sub_decode<true>(d, j, st, sub_stuffing(d, j), d.sub_off[j], s_lut, s_qt, s_zz);
}

// dcabs: [n luma blocks] running sum of the DC differences (decode order).  One lane per luma block.
__global__ __launch_bounds__(256) void jpeg_dc_apply_kernel(const JpegImageDesc* __restrict__ descs, const int32_t* const* dcabs) {
  const JpegImageDesc d = descs[blockIdx.y];
  const int yblocks = d.hs0 * d.vs0;
  const long long n_luma = (long long)d.total_mcus * yblocks;
  const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
  if (o >= n_luma) return;
  const int mcu = (int)(o / yblocks), b = (int)(o - (long long)mcu * yblocks);
  const int my = mcu / d.mcus_x, mx = mcu - my * d.mcus_x;
  const int bx = mx * d.hs0 + (b % d.hs0), by = my * d.vs0 + (b / d.hs0);
  const int v = dcabs[blockIdx.y][o] * (int)d.tables->qt[0][0];
  d.coef[((long long)by * d.Wb + bx) * 64] = (int16_t)max(-32768, min(32767, v));
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

// Restart-less streams, first half: synchronisation (kernels a and b).  max_sub: largest n_sub of the batch.
extern "C" int dlamd_jpeg_sync(const void* descs, int n, int max_sub, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n > 65535 || max_sub <= 0) return -1;
  const int chunks = (max_sub + JPEG_SUB_CHUNK - 1) / JPEG_SUB_CHUNK;
  hipLaunchKernelGGL(jpeg_sync_kernel, dim3(chunks, n), dim3(JPEG_SUB_CHUNK), 0, stream, (const JpegImageDesc*)descs);
  if (chunks > 1) hipLaunchKernelGGL(jpeg_sync_seam_kernel, dim3(n), dim3(256), 0, stream, (const JpegImageDesc*)descs);
  return DLAMD_CHECK_LAUNCH();
}

// Second half, after the caller has turned sub_count into sub_off: coefficient decode (kernel d).
extern "C" int dlamd_jpeg_sync_decode(const void* descs, int n, int max_sub, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n > 65535 || max_sub <= 0) return -1;
  hipLaunchKernelGGL(jpeg_sync_decode_kernel, dim3((max_sub + 127) / 128, n), dim3(128), 0, stream, (const JpegImageDesc*)descs);
  return DLAMD_CHECK_LAUNCH();
}

// Last part, after the caller's prefix sum over dcdiff: absolute DC values into the coefficient planes, then the IDCT.
// dcabs: device array of n pointers.  max_luma / max_blocks: maxima over the batch.
extern "C" int dlamd_jpeg_sync_finish(const void* descs, const void* dcabs, int n, long long max_luma, long long max_blocks,
                                      hipStream_t stream) {
  if (n <= 0) return 0;
  if (n > 65535 || max_luma <= 0 || max_blocks <= 0) return -1;
  const long long g1 = (max_luma + 255) / 256, g2 = (max_blocks + 31) / 32;
  if (g1 > 0x7fffffffLL || g2 > 0x7fffffffLL) return -2;
  hipLaunchKernelGGL(jpeg_dc_apply_kernel, dim3((unsigned)g1, n), dim3(256), 0, stream, (const JpegImageDesc*)descs,
                     (const int32_t* const*)dcabs);
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)g2, n), dim3(256), 0, stream, (const JpegImageDesc*)descs);
  return DLAMD_CHECK_LAUNCH();
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
