// DB head tail, fused: deconv 2x2/s2 (64 -> 64, folded BN, ReLU) + deconv 2x2/s2 (64 -> 1) + sigmoid + threshold
// in one kernel.  Unfused, the first deconv writes a [B, H/2, W/2, 64] bf16 map (660 MB for a 25-page
// batch) that the second one reads straight back; both launches are pure memory traffic (44 - 57 TFLOP/s in
// bench/conv_layer_breakdown.py).  Fused, only the probability map and the bit mask leave the chip.
//
// Per 16 quarter-resolution pixels (one MFMA column group), all inside one wave:
//   GEMM1  D1[256 x 16] = W2[256 x 64] . X[64 x 16]     rows n = (g, c'): g = 2x2 sub-pixel, c' = mid channel
//   v      = bf16(relu(D1 + b2))                         exactly what the unfused path stores
//   GEMM2  D2[16 x 16]  = A2[16 x 256] . v               rows (q, r): q = sub-pixel, r = its 2x2 sub-sub-pixel;
//                                                        A2 is block diagonal: A2[(q,r),(g,c')] = [q == g] W3[r][c']
//   prob   = bf16(sigmoid(D2 + b3)), mask = prob > thr   -> 4 x 64 full-resolution pixels
// The K order of GEMM2 is chosen so that its B fragment is lane-local: K-step s takes rows 4fq..4fq+3 of D1 tiles
// 2s and 2s+1, which are exactly the 8 accumulator values the lane already holds (the host permutes A2 to match).
// Weights of both GEMMs stay in registers; waves stride over the pixel groups (persistent), the next group's input
// is prefetched during the current group's math, results go out through a per-wave LDS slab as full 16 / 8-byte rows.
#include "common.h"

namespace {

struct DbHeadArgs {
  const bf16_t* h;        // [B, Hq, Wq, 64]
  const bf16_t* w2;       // [256, 64] rows (g, c')
  const float* b2;        // [256]
  const bf16_t* a2frag;   // [8][64][8] A fragments of GEMM2 (host-permuted, see ops/conv.py: db_head_a2_fragments)
  const float* b3;        // [4]
  bf16_t* prob;           // [B, 4Hq, 4Wq]
  uint8_t* mask;          // [B, 4Hq, 4Wq]
  int B, Hq, Wq, gx;      // gx = ceil(Wq / 16)
  long long groups;
  float thr;
};

__global__ __launch_bounds__(256) void db_head_fused_kernel(DbHeadArgs p) {
  __shared__ __attribute__((aligned(16))) char slab[4][4 * 128 + 4 * 64];   // per wave: prob 4 x 64 bf16, mask 4 x 64 u8
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, fq = lane >> 4;

  bf16x8 w2f[16][2];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s)
      w2f[t][s] = *reinterpret_cast<const bf16x8*>(p.w2 + (long long)(t * 16 + col) * 64 + s * 32 + fq * 8);
  bf16x8 a2f[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) a2f[s] = *reinterpret_cast<const bf16x8*>(p.a2frag + ((long long)s * 64 + lane) * 8);
  const f32x4 b3v = *reinterpret_cast<const f32x4*>(p.b3);

  char* my = slab[wave];
  const long long nw = (long long)gridDim.x * 4;
  long long gi = (long long)blockIdx.x * 4 + wave;

  auto load_group = [&](long long g, u32x4 (&xf)[2]) {
    xf[0] = u32x4{0u, 0u, 0u, 0u};
    xf[1] = xf[0];
    if (g < p.groups) {
      const int xg = (int)(g % p.gx);
      const long long row = g / p.gx;               // b * Hq + y
      const int x = xg * 16 + col;
      if (x < p.Wq) {
        const bf16_t* src = p.h + (row * p.Wq + x) * 64 + fq * 8;
        xf[0] = *reinterpret_cast<const u32x4*>(src);
        xf[1] = *reinterpret_cast<const u32x4*>(src + 32);
      }
    }
  };

  u32x4 cur[2], nxt[2];
  load_group(gi, cur);
  for (; gi < p.groups; gi += nw) {
    load_group(gi + nw, nxt);
    const int xg = (int)(gi % p.gx);
    const long long row = gi / p.gx;
    const int y = (int)(row % p.Hq), b = (int)(row / p.Hq);
    const int x0 = xg * 16;

    f32x4 d2 = b3v;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      f32x4 d1[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * s + u;
        d1[u] = *reinterpret_cast<const f32x4*>(p.b2 + t * 16 + fq * 4);
        d1[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][0], as_bf16x8(cur[0]), d1[u], 0, 0, 0);
        d1[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[t][1], as_bf16x8(cur[1]), d1[u], 0, 0, 0);
      }
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = (bf16_t)fmaxf(d1[0][j], 0.f);
        v[4 + j] = (bf16_t)fmaxf(d1[1][j], 0.f);
      }
      d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2f[s], v, d2, 0, 0, 0);
    }
    // lane (col, fq): quarter pixel x0 + col, sub-pixel q = fq, sub-sub-pixels r = 0..3
    const int qy = fq >> 1, qx = fq & 1;
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
      const bf16_t p0 = (bf16_t)(1.f / (1.f + __expf(-d2[ry * 2])));
      const bf16_t p1 = (bf16_t)(1.f / (1.f + __expf(-d2[ry * 2 + 1])));
      const int yl = 2 * qy + ry, xl = 4 * col + 2 * qx;
      *reinterpret_cast<bf16x2*>(&my[yl * 128 + xl * 2]) = bf16x2{p0, p1};
      // threshold the ROUNDED value so mask and stored probability always agree
      *reinterpret_cast<uint16_t*>(&my[512 + yl * 64 + xl]) =
          uint16_t((float)p0 > p.thr ? 1 : 0) | uint16_t(((float)p1 > p.thr ? 1 : 0) << 8);
    }
    const int npx = min(16, p.Wq - x0);              // even (host contract: Wq % 2 == 0)
    const long long W4 = 4ll * p.Wq;
    const long long obase = ((long long)b * 4 * p.Hq + 4 * y) * W4 + 4 * x0;
    if (lane < 32) {
      const int r = lane >> 3, ch = lane & 7;        // 16-byte chunk = 8 full-resolution pixels = 2 quarter pixels
      const u32x4 v = *reinterpret_cast<const u32x4*>(&my[r * 128 + ch * 16]);
      if (ch * 2 < npx) *reinterpret_cast<u32x4*>(p.prob + obase + r * W4 + ch * 8) = v;
    } else {
      const int l2 = lane - 32, r = l2 >> 3, ch = l2 & 7;   // 8-byte chunk of the mask = the same 8 pixels
      const u32x2 v = *reinterpret_cast<const u32x2*>(&my[512 + r * 64 + ch * 8]);
      if (ch * 2 < npx) *reinterpret_cast<u32x2*>(p.mask + obase + r * W4 + ch * 8) = v;
    }
    cur[0] = nxt[0];
    cur[1] = nxt[1];
  }
}

// Block head tail: the [N, 8] logits of the quarter-scale block head -> probability map (bf16) and bit mask of the
// first one or two channels (figures, formulas), each as its own dense [N] plane for the connected-component pass.
// One 16-byte load per pixel; a wave writes 128 contiguous bytes of probabilities and 64 of mask per channel.
struct BlockSplitArgs {
  const bf16_t* logits;   // [N, 8]
  bf16_t* prob[2];        // [N] each
  uint8_t* mask[2];
  long long n;
  int nch;
};

__global__ __launch_bounds__(256) void block_head_split_kernel(BlockSplitArgs p) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
    const bf16x8 v = as_bf16x8(*(const u32x4*)(p.logits + i * 8));
    const float l0 = (float)v[0], l1 = (float)v[1];
    p.prob[0][i] = (bf16_t)(1.f / (1.f + __expf(-l0)));
    p.mask[0][i] = l0 > 0.f ? 1 : 0;
    if (p.nch > 1) {
      p.prob[1][i] = (bf16_t)(1.f / (1.f + __expf(-l1)));
      p.mask[1][i] = l1 > 0.f ? 1 : 0;
    }
  }
}

}  // namespace

// Host contract: logits [N, 8] bf16 contiguous and 16-byte aligned; prob planes [N] bf16, mask planes [N] u8; the second
// pair may be null (one channel).
extern "C" int dlamd_block_head_split(const void* logits, void* prob0, void* mask0, void* prob1, void* mask1, long long n,
                                      hipStream_t stream) {
  if (!logits || !prob0 || !mask0 || n <= 0) return -1;
  if ((prob1 == nullptr) != (mask1 == nullptr)) return -2;
  if (((uintptr_t)logits & 15) != 0) return -3;
  BlockSplitArgs a;
  a.logits = (const bf16_t*)logits;
  a.prob[0] = (bf16_t*)prob0; a.mask[0] = (uint8_t*)mask0;
  a.prob[1] = (bf16_t*)prob1; a.mask[1] = (uint8_t*)mask1;
  a.n = n;
  a.nch = prob1 ? 2 : 1;
  long long wgs = (n + 255) / 256;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(block_head_split_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}

// Host contract: h [B,Hq,Wq,64] bf16 contiguous, Wq even, prob / mask [B,4Hq,4Wq] contiguous; all pointers 16-byte aligned.
extern "C" int dlamd_db_head_fused(const void* h, const void* w2, const float* b2, const void* a2frag, const float* b3,
                                   void* prob, void* mask, int B, int Hq, int Wq, float thr, hipStream_t stream) {
  if (!h || !w2 || !b2 || !a2frag || !b3 || !prob || !mask) return -1;
  if (B <= 0 || Hq <= 0 || Wq <= 0 || (Wq & 1)) return -2;
  DbHeadArgs a;
  a.h = (const bf16_t*)h; a.w2 = (const bf16_t*)w2; a.b2 = b2; a.a2frag = (const bf16_t*)a2frag; a.b3 = b3;
  a.prob = (bf16_t*)prob; a.mask = (uint8_t*)mask;
  a.B = B; a.Hq = Hq; a.Wq = Wq; a.gx = (Wq + 15) / 16;
  a.groups = (long long)B * Hq * a.gx;
  a.thr = thr;
  if (((4ll * B * Hq) * (4ll * Wq)) > 0x7fffffff0ll) return -3;
  long long wgs = (a.groups + 3) / 4;
  if (wgs > 512) wgs = 512;                          // 2 workgroups per CU; waves stride over the groups
  hipLaunchKernelGGL(db_head_fused_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -9;
}
