// Row-wise prefix sums of int32 matrices [rows][len] (the JPEG decoder's block counts per subsequence and DC
// differences per block: 36 rows x 136 k elements per batch of pages).  Three small launches: per-chunk sums, a scan
// of the chunk sums of each row, and the chunk-local scan with its offset added - every row is spread over
// len / 1024 workgroups instead of one.
#include "common.h"

static constexpr int SCAN_CHUNK = 1024;   // elements per workgroup (256 threads x 4)

__device__ __forceinline__ int block_exclusive_scan_256(int v, int* lds, int& total) {
  // wave scan, then the four wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int t = lds[w];
    if (w < wave) base += t;
  }
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const int* __restrict__ in, int* __restrict__ sums, int len,
                                                              int nchunks) {
  __shared__ int lds[4];
  const int row = blockIdx.y, chunk = blockIdx.x;
  const int* r = in + (long long)row * len;
  const int e0 = chunk * SCAN_CHUNK + threadIdx.x * 4;
  int v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (e0 + k < len) v += r[e0 + k];
  int total;
  block_exclusive_scan_256(v, lds, total);
  if (threadIdx.x == 0) sums[(long long)row * nchunks + chunk] = total;
}

// one workgroup per row: chunk sums -> exclusive chunk offsets, in place
__global__ __launch_bounds__(256) void scan_chunk_offsets_kernel(int* __restrict__ sums, int nchunks) {
  __shared__ int lds[4];
  int* s = sums + (long long)blockIdx.x * nchunks;
  int carry = 0;
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    const int c = c0 + threadIdx.x;
    const int v = c < nchunks ? s[c] : 0;
    int total;
    const int ex = block_exclusive_scan_256(v, lds, total);
    if (c < nchunks) s[c] = carry + ex;
    carry += total;
  }
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const int* __restrict__ in, const int* __restrict__ offs,
                                                         int* __restrict__ out, int len, int nchunks, int exclusive) {
  __shared__ int lds[4];
  const int row = blockIdx.y, chunk = blockIdx.x;
  const int* r = in + (long long)row * len;
  int* o = out + (long long)row * len;
  const int e0 = chunk * SCAN_CHUNK + threadIdx.x * 4;
  int x[4];
  int v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x[k] = e0 + k < len ? r[e0 + k] : 0;
    v += x[k];
  }
  int total;
  int run = block_exclusive_scan_256(v, lds, total) + offs[(long long)row * nchunks + chunk];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (exclusive) { if (e0 + k < len) o[e0 + k] = run; run += x[k]; }
    else { run += x[k]; if (e0 + k < len) o[e0 + k] = run; }
  }
}

// in, out: [rows][len] int32 (may not alias unless identical); scratch: at least rows * ceil(len / 1024) int32
extern "C" int dlamd_row_cumsum_i32(const void* in, void* out, void* scratch, long long scratch_elems, int rows, int len,
                                    int exclusive, hipStream_t stream) {
  if (rows <= 0 || len <= 0) return 0;
  const int nchunks = (len + SCAN_CHUNK - 1) / SCAN_CHUNK;
  if (rows > 65535 || (long long)rows * nchunks > scratch_elems) return -1;
  hipLaunchKernelGGL(scan_chunk_sums_kernel, dim3(nchunks, rows), dim3(256), 0, stream, (const int*)in, (int*)scratch, len,
                     nchunks);
  hipLaunchKernelGGL(scan_chunk_offsets_kernel, dim3(rows), dim3(256), 0, stream, (int*)scratch, nchunks);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nchunks, rows), dim3(256), 0, stream, (const int*)in, (const int*)scratch,
                     (int*)out, len, nchunks, exclusive);
  return DLAMD_CHECK_LAUNCH();
}
