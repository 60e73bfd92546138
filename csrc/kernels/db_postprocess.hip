// DB (differentiable binarisation) post-processing on the GPU.
//
// Replaces the OpenCV findContours + pyclipper + shapely chain of the
// reference's dependency stack with four small kernels that never move the
// full-resolution mask to the host:
//   1. ccl_init     parent[p] = p for foreground pixels, -1 for background
//   2. ccl_merge    union-find over left / up / up-left / up-right neighbours
//                   (8-connectivity), lock-free with atomicMin on roots
//   3. ccl_assign   path-compress and give every root a compact component id
//      ccl_stats    reduce per-component bbox / probability sum / pixel count.
//                   Lanes of a wave that hit the same component are combined
//                   in-register first (text blobs are wide, so a wave64 almost
//                   always sits inside one component) -> one atomic set per
//                   wave instead of 64 contended ones.
//   4. box_finalize score filter, orientation from the component's second moments (a skewed scan gives slanted
//                   text lines: the box becomes the minimum-area rectangle around the blob, reported as an upright
//                   rectangle plus the angle to turn it by about its centre), closed-form rectangle "unclip"
//                   (d = area * ratio / perimeter), rescale to page pixels.
// Output is a small [B, max_boxes, 8] float table + per-page counts.
#include "common.h"

#define MAXC_DEFAULT 4096

// Page-local indices are used inside parent[] (index within the page).
// Every foreground pixel starts under the first pixel of its horizontal run inside its 64-pixel wave segment
// (segments are 64 consecutive flattened positions; a run never continues across a row start).  One ballot per
// wave pre-merges what would otherwise be ~one atomic union per pixel: text blobs are wide, so this removes almost
// all horizontal unions and keeps the union-find trees shallow.
__global__ void ccl_init_local_kernel(const uint8_t* __restrict__ mask, int* __restrict__ parent, int hw, int B, int W) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  int b = blockIdx.y;
  const bool in = p < hw;
  long long g = (long long)b * hw + (in ? p : 0);
  const bool fg = in && mask[g] != 0;
  const int lane = threadIdx.x & 63;
  const unsigned long long fgm = __ballot(fg);
  const unsigned long long rowstart = __ballot(in && (p % W) == 0);
  const unsigned long long starts = fgm & (~(fgm << 1) | rowstart);
  if (!in) return;
  int par = -1;
  if (fg) {
    const unsigned long long below = starts & (~0ull >> (63 - lane));     // run starts at lanes <= lane
    par = p - (lane - (63 - __builtin_clzll(below)));
  }
  parent[g] = par;
}

__device__ __forceinline__ int uf_find(int* parent, int x) {
  int r = x;
  while (true) {
    int pr = parent[r];
    if (pr == r) break;
    r = pr;
  }
  return r;
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) { int t = a; a = b; b = t; }   // a > b: attach a under b
    int old = atomicMin(&parent[a], b);
    if (old == a) return;
    a = old;  // someone else changed it; retry with the new link
  }
}

__global__ void ccl_merge_kernel(const uint8_t* __restrict__ mask, int* __restrict__ parent_all, int H, int W) {
  int x = blockIdx.x * 64 + (threadIdx.x & 63);
  int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  int b = blockIdx.z;
  if (x >= W || y >= H) return;
  const long long base = (long long)b * H * W;
  const uint8_t* m = mask + base;
  int* parent = parent_all + base;
  int p = y * W + x;
  if (!m[p]) return;
  // horizontal: runs were pre-merged inside 64-pixel segments (ccl_init_local_kernel); only segment seams remain
  if (x > 0 && (p & 63) == 0 && m[p - 1]) uf_union(parent, p, p - 1);
  if (y > 0) {
    if (m[p - W]) {
      // vertical: one union per overlap of two runs (its leftmost pixel), the rest follows from the rows' own runs
      if (x == 0 || !m[p - 1] || !m[p - W - 1]) uf_union(parent, p, p - W);
    } else {
      // only needed when the pixel straight above is background
      if (x > 0 && m[p - W - 1]) uf_union(parent, p, p - W - 1);
      if (x + 1 < W && m[p - W + 1]) uf_union(parent, p, p - W + 1);
    }
  }
}

// Path-compress every foreground pixel to its root and give each root a compact id.
__global__ void ccl_assign_kernel(int* __restrict__ parent_all, int* __restrict__ comp_id_all,
                                  int* __restrict__ ncomp, int H, int W, int maxc) {
  int x = blockIdx.x * 64 + (threadIdx.x & 63);
  int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  int b = blockIdx.z;
  if (x >= W || y >= H) return;
  const long long base = (long long)b * H * W;
  int* parent = parent_all + base;
  int p = y * W + x;
  if (parent[p] < 0) return;
  int root = uf_find(parent, p);
  if (root == p) {
    int id = atomicAdd(&ncomp[b], 1);
    comp_id_all[base + p] = id < maxc ? id : maxc;
  } else {
    parent[p] = root;  // benign race: every value ever stored is an ancestor of p
  }
}

// stats layout per component, STAT_STRIDE int32 words: [xmin, ymin, xmax, ymax, count, sum(prob) as float bits,
// then five 64-bit sums over the component's pixels: x, y, x*x, y*y, x*y] (exact integers: box angles do not depend
// on the order the atomics land in).
//
// Contention control, three levels deep (a text blob is wide and many rows tall, so naive per-pixel
// atomics all hit the same six words):
//   1. lanes of a wave that see the same component are combined in registers (shuffles),
//   2. the wave leaders of a 64 x 16 pixel block merge into a small LDS hash table (LDS atomics),
//   3. one global atomic set per (block, component) is issued at the end.
#define STAT_SLOTS 64
#define STAT_ROWS 16
#define STAT_STRIDE 16

// n pixels of one component with block-relative sums (x' = x - bx0, y' = y - by0) -> absolute sums, added to global
__device__ __forceinline__ void moments_global_add(int* __restrict__ s, long long n, long long sx, long long sy, long long sxx,
                                                   long long syy, long long sxy, long long bx0, long long by0) {
  unsigned long long* m = reinterpret_cast<unsigned long long*>(s + 6);
  atomicAdd(&m[0], (unsigned long long)(sx + n * bx0));
  atomicAdd(&m[1], (unsigned long long)(sy + n * by0));
  atomicAdd(&m[2], (unsigned long long)(sxx + 2 * bx0 * sx + n * bx0 * bx0));
  atomicAdd(&m[3], (unsigned long long)(syy + 2 * by0 * sy + n * by0 * by0));
  atomicAdd(&m[4], (unsigned long long)(sxy + bx0 * sy + by0 * sx + n * bx0 * by0));
}

__device__ __forceinline__ void stats_global_add(int* __restrict__ s, int xmin, int ymin, int xmax, int ymax, int cnt,
                                                 float ps) {
  atomicMin(&s[0], xmin);
  atomicMin(&s[1], ymin);
  atomicMax(&s[2], xmax);
  atomicMax(&s[3], ymax);
  atomicAdd(&s[4], cnt);
  atomicAdd(reinterpret_cast<float*>(&s[5]), ps);
}

__global__ __launch_bounds__(256) void ccl_stats_kernel(const bf16_t* __restrict__ prob, const int* __restrict__ parent_all,
                                                        const int* __restrict__ comp_id_all, int* __restrict__ stats,
                                                        int H, int W, int maxc) {
  __shared__ int t_cid[STAT_SLOTS];
  __shared__ int t_xmin[STAT_SLOTS], t_ymin[STAT_SLOTS], t_xmax[STAT_SLOTS], t_ymax[STAT_SLOTS], t_cnt[STAT_SLOTS];
  __shared__ float t_ps[STAT_SLOTS];
  __shared__ int t_sx[STAT_SLOTS], t_sy[STAT_SLOTS], t_sxx[STAT_SLOTS], t_syy[STAT_SLOTS], t_sxy[STAT_SLOTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < STAT_SLOTS) {
    t_cid[tid] = -1; t_xmin[tid] = 0x7fffffff; t_ymin[tid] = 0x7fffffff; t_xmax[tid] = -1; t_ymax[tid] = -1;
    t_cnt[tid] = 0; t_ps[tid] = 0.f;
    t_sx[tid] = 0; t_sy[tid] = 0; t_sxx[tid] = 0; t_syy[tid] = 0; t_sxy[tid] = 0;
  }
  __syncthreads();
  const int x = blockIdx.x * 64 + lane;
  const int b = blockIdx.z;
  const long long base = (long long)b * H * W;
  const int* parent = parent_all + base;
  for (int r = 0; r < STAT_ROWS / 4; ++r) {
    const int y = blockIdx.y * STAT_ROWS + r * 4 + wave;
    const bool inb = (x < W && y < H);
    const int p = y * W + x;
    int cid = -1;
    float pv = 0.f;
    if (inb) {
      int root = parent[p];
      if (root >= 0) {
        // after ccl_assign every parent is either the root or one hop from it
        int r2 = parent[root];
        if (r2 != root) { root = r2; int r3 = parent[root]; while (r3 != root) { root = r3; r3 = parent[root]; } }
        cid = comp_id_all[base + root];
        pv = (float)prob[base + p];
      }
    }
    const bool active = (cid >= 0 && cid < maxc);
    unsigned long long todo = __ballot(active);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int lc = __shfl(cid, leader, 64);
      const bool same = active && (cid == lc);
      const unsigned long long grp = __ballot(same);
      int vxmin = same ? x : 0x7fffffff, vxmax = same ? x : -1;
      int vcnt = same ? 1 : 0;
      float vps = same ? pv : 0.f;
      int vsx = same ? lane : 0, vsxx = same ? lane * lane : 0;        // x relative to the block's first column
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        vxmin = min(vxmin, __shfl_xor(vxmin, o, 64));
        vxmax = max(vxmax, __shfl_xor(vxmax, o, 64));
        vcnt += __shfl_xor(vcnt, o, 64);
        vps += __shfl_xor(vps, o, 64);
        vsx += __shfl_xor(vsx, o, 64);
        vsxx += __shfl_xor(vsxx, o, 64);
      }
      if (lane == leader) {
        // insert / merge into the block's LDS table (open addressing, linear probing)
        int slot = (int)(((unsigned)lc * 2654435761u) >> 26) & (STAT_SLOTS - 1);
        bool done = false;
        for (int probe = 0; probe < STAT_SLOTS && !done; ++probe) {
          const int cur = atomicCAS(&t_cid[slot], -1, lc);
          if (cur == -1 || cur == lc) {
            atomicMin(&t_xmin[slot], vxmin);
            atomicMax(&t_xmax[slot], vxmax);
            atomicMin(&t_ymin[slot], y);
            atomicMax(&t_ymax[slot], y);
            atomicAdd(&t_cnt[slot], vcnt);
            atomicAdd(&t_ps[slot], vps);
            const int yr = y - blockIdx.y * STAT_ROWS;                   // a wave covers one row: y is uniform
            atomicAdd(&t_sx[slot], vsx);
            atomicAdd(&t_sxx[slot], vsxx);
            atomicAdd(&t_sy[slot], vcnt * yr);
            atomicAdd(&t_syy[slot], vcnt * yr * yr);
            atomicAdd(&t_sxy[slot], vsx * yr);
            done = true;
          }
          slot = (slot + 1) & (STAT_SLOTS - 1);
        }
        // more than STAT_SLOTS distinct components inside one 64 x 16 block (salt-and-pepper masks): go direct
        if (!done) {
          int* sg = stats + ((long long)b * maxc + lc) * STAT_STRIDE;
          stats_global_add(sg, vxmin, y, vxmax, y, vcnt, vps);
          moments_global_add(sg, vcnt, vsx, 0, vsxx, 0, 0, (long long)blockIdx.x * 64, y);
        }
      }
      todo &= ~grp;
    }
  }
  __syncthreads();
  if (tid < STAT_SLOTS && t_cid[tid] >= 0) {
    int* sg = stats + ((long long)b * maxc + t_cid[tid]) * STAT_STRIDE;
    stats_global_add(sg, t_xmin[tid], t_ymin[tid], t_xmax[tid], t_ymax[tid], t_cnt[tid], t_ps[tid]);
    moments_global_add(sg, t_cnt[tid], t_sx[tid], t_sy[tid], t_sxx[tid], t_syy[tid], t_sxy[tid],
                       (long long)blockIdx.x * 64, (long long)blockIdx.y * STAT_ROWS);
  }
}

// Page skew: every wide component votes for the angle of its principal axis with its width; the page's angle is the
// mode of that histogram.  One angle per page is far more precise than any single blob's second moments (a text blob
// is not a clean rectangle: ascenders, descenders and ragged ends pull its axis by tenths of a degree, which over a
// 2000-pixel line is a dozen pixels of drift in the crop).
#define ANG_BINS 401            // 0.1 degree bins over +-20 degrees
#define ANG_BIN_RAD 0.0017453292f

__global__ void stats_init_kernel(int* __restrict__ stats, int* __restrict__ ncomp, int* __restrict__ nbox, int n,
                                  int B, int* __restrict__ hist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { ncomp[i] = 0; nbox[i] = 0; }
  if (hist && i < B * ANG_BINS) hist[i] = 0;
  if (i >= n) return;
  int* s = stats + (long long)i * STAT_STRIDE;
  s[0] = 0x7fffffff; s[1] = 0x7fffffff; s[2] = -1; s[3] = -1; s[4] = 0; s[5] = 0;
#pragma unroll
  for (int k = 6; k < STAT_STRIDE; ++k) s[k] = 0;
}

// principal-axis angle of component `s` (radians, det-map frame) and the eigenvalues of its covariance
__device__ __forceinline__ double component_axis(const int* __restrict__ s, int cnt, double& l_major, double& l_minor,
                                                 double& mx, double& my) {
  const long long* m = reinterpret_cast<const long long*>(s + 6);
  const double n = (double)cnt;
  mx = (double)m[0] / n;
  my = (double)m[1] / n;
  const double u20 = (double)m[2] / n - mx * mx, u02 = (double)m[3] / n - my * my, u11 = (double)m[4] / n - mx * my;
  const double r = sqrt(0.25 * (u20 - u02) * (u20 - u02) + u11 * u11);
  l_major = 0.5 * (u20 + u02) + r;
  l_minor = fmax(0.5 * (u20 + u02) - r, 0.0);
  return 0.5 * atan2(2.0 * u11, u20 - u02);
}

__global__ void angle_vote_kernel(const int* __restrict__ stats, const int* __restrict__ ncomp, int* __restrict__ hist,
                                  int maxc, float box_thresh, int min_size, float max_angle) {
  int cidx = blockIdx.x * blockDim.x + threadIdx.x;
  int b = blockIdx.y;
  if (cidx >= min(ncomp[b], maxc)) return;
  const int* s = stats + ((long long)b * maxc + cidx) * STAT_STRIDE;
  const int cnt = s[4];
  if (cnt <= 0) return;
  const float w = (float)(s[2] + 1 - s[0]), h = (float)(s[3] + 1 - s[1]);
  if (fminf(w, h) < (float)min_size || w < 3.f * h) return;
  if (__builtin_bit_cast(float, s[5]) / (float)cnt < box_thresh) return;
  double l1, l2, mx, my;
  const double th = component_axis(s, cnt, l1, l2, mx, my);
  if (fabs(th) > (double)max_angle || l1 < 64.0 * (l2 + 1.0 / 12.0)) return;   // length >= 8 x thickness
  const int bin = (int)lrint(th / (double)ANG_BIN_RAD) + ANG_BINS / 2;
  if (bin >= 0 && bin < ANG_BINS) atomicAdd(&hist[b * ANG_BINS + bin], (int)w);
}

// one wave per page: mode of the 3-bin-smoothed histogram, refined by the centroid of those three bins
__global__ __launch_bounds__(64) void angle_mode_kernel(const int* __restrict__ hist, float* __restrict__ page_angle) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int* h = hist + b * ANG_BINS;
  long long best = -1;
  int best_bin = ANG_BINS / 2;
  for (int i = lane; i < ANG_BINS; i += 64) {
    const long long v = (long long)h[i] + (i > 0 ? h[i - 1] : 0) + (i + 1 < ANG_BINS ? h[i + 1] : 0);
    if (v > best) { best = v; best_bin = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const long long ov = __shfl_xor(best, o, 64);
    const int ob = __shfl_xor(best_bin, o, 64);
    if (ov > best || (ov == best && ob < best_bin)) { best = ov; best_bin = ob; }
  }
  if (lane == 0) {
    float a = 0.f;
    if (best > 0) {
      const float c0 = best_bin > 0 ? (float)h[best_bin - 1] : 0.f, c1 = (float)h[best_bin],
                  c2 = best_bin + 1 < ANG_BINS ? (float)h[best_bin + 1] : 0.f;
      a = ((float)(best_bin - ANG_BINS / 2) + (c2 - c0) / (c0 + c1 + c2)) * ANG_BIN_RAD;
    }
    page_angle[b] = a;
  }
}

// page_scale: [B, 2] float (sx, sy): multiply det-map coords to get page pixel coords.
// valid: [B, 2] int (valid_w, valid_h) of the det map (content area, rest is bucket padding)
// boxes: [B, max_boxes, 8] float: x0, y0, x1, y1 (page pixels), score, pixel count, angle, 0.  With angle != 0 the
// text line is the rectangle x0..y1 turned by `angle` radians about its centre (positive = the line runs down to the
// right in image coordinates).
// max_angle: largest |angle| reported (radians); wider components whose principal axis is tilted less than this get
// an oriented box, everything else (and max_angle = 0) the upright bounding box.
__global__ void box_finalize_kernel(const int* __restrict__ stats, const int* __restrict__ ncomp,
                                    const float* __restrict__ page_scale, const int* __restrict__ valid,
                                    float* __restrict__ boxes, int* __restrict__ nbox, int maxc, int max_boxes,
                                    float box_thresh, float unclip_ratio, int min_size, float max_angle,
                                    const float* __restrict__ page_angle, int H, int W) {
  int cidx = blockIdx.x * blockDim.x + threadIdx.x;
  int b = blockIdx.y;
  int nc = min(ncomp[b], maxc);
  if (cidx >= nc) return;
  const int* s = stats + ((long long)b * maxc + cidx) * STAT_STRIDE;
  int cnt = s[4];
  if (cnt <= 0) return;
  float score = __builtin_bit_cast(float, s[5]) / (float)cnt;
  float x0 = (float)s[0], y0 = (float)s[1], x1 = (float)(s[2] + 1), y1 = (float)(s[3] + 1);
  float w = x1 - x0, h = y1 - y0;
  if (fminf(w, h) < (float)min_size) return;
  if (score < box_thresh) return;
  const float sx = page_scale ? page_scale[b * 2] : 1.f, sy = page_scale ? page_scale[b * 2 + 1] : 1.f;
  // orientation: the page's skew (angle_mode_kernel) for every wide component; a wide component whose own principal
  // axis disagrees with it by more than 3 degrees (a rotated inset) keeps its own
  float angle = 0.f;
  double l_major = 0.0, l_minor = 0.0, mx = 0.0, my = 0.0;
  if (max_angle > 0.f && w >= 3.f * h) {
    const double th = component_axis(s, cnt, l_major, l_minor, mx, my);
    const float pa = page_angle ? page_angle[b] : (float)th;
    const bool own_ok = fabs(th) <= (double)max_angle && l_major >= 16.0 * (l_minor + 1.0 / 12.0);
    float a = pa;
    if (own_ok && l_major >= 64.0 * (l_minor + 1.0 / 12.0) && fabs(th - (double)pa) > 0.0524) a = (float)th;
    if (fabsf(a) >= 0.004f && fabsf(a) <= max_angle) angle = a;
  }
  if (angle == 0.f) {
    float d = (w * h) * unclip_ratio / (2.f * (w + h));
    int vw = valid ? valid[b * 2] : W, vh = valid ? valid[b * 2 + 1] : H;
    x0 = fmaxf(x0 - d, 0.f); y0 = fmaxf(y0 - d, 0.f);
    x1 = fminf(x1 + d, (float)vw); y1 = fminf(y1 + d, (float)vh);
    if (x1 - x0 < (float)(min_size + 2) || y1 - y0 < (float)(min_size + 2)) return;
    int slot = atomicAdd(&nbox[b], 1);
    if (slot >= max_boxes) return;
    float* o = boxes + ((long long)b * max_boxes + slot) * 8;
    o[0] = x0 * sx; o[1] = y0 * sy; o[2] = x1 * sx; o[3] = y1 * sy; o[4] = score; o[5] = (float)cnt; o[6] = 0.f; o[7] = 0.f;
    return;
  }
  // A rectangle L x T turned by `angle` has the upright bounding box w = L c + T s, h = L s + T c (c = cos, s = |sin|):
  // solve for L and T (exact for a filled rectangle, and driven by the outermost pixels, so ragged line ends do not
  // shorten the box), bounded below by the extents the second moments imply (variance of a uniform bar = length^2 / 12).
  const float c = cosf(angle), sg = sinf(angle), sa = fabsf(sg);
  const float c2 = c * c - sa * sa;
  float L = (w * c - h * sa) / c2, T = (h * c - w * sa) / c2;
  L = fminf(fmaxf(L, 0.8f * sqrtf(12.f * (float)l_major)), sqrtf(w * w + h * h));
  T = fminf(fmaxf(T, fmaxf(1.f, 0.8f * sqrtf(12.f * (float)l_minor + 1.f))), h);
  if (T < (float)min_size) return;
  const float d = (L * T) * unclip_ratio / (2.f * (L + T));
  L += 2.f * d;
  T += 2.f * d;
  if (L < (float)(min_size + 2) || T < (float)(min_size + 2)) return;
  // to page pixels (sx and sy may differ slightly): directions scale component-wise
  const float ux = c * sx, uy = sg * sy, vx = -sg * sx, vy = c * sy;
  const float Lp = L * sqrtf(ux * ux + uy * uy), Tp = T * sqrtf(vx * vx + vy * vy);
  const float ap = atan2f(uy, ux);
  const float cxp = ((float)mx + 0.5f) * sx, cyp = ((float)my + 0.5f) * sy;
  int slot = atomicAdd(&nbox[b], 1);
  if (slot >= max_boxes) return;
  float* o = boxes + ((long long)b * max_boxes + slot) * 8;
  o[0] = cxp - 0.5f * Lp; o[1] = cyp - 0.5f * Tp; o[2] = cxp + 0.5f * Lp; o[3] = cyp + 0.5f * Tp;
  o[4] = score; o[5] = (float)cnt; o[6] = ap; o[7] = 0.f;
}

// mask: [B,H,W] u8, prob: [B,H,W] bf16.  Workspaces: parent, comp_id: [B,H,W] int32;
// stats: [B,maxc,16] int32 (8-byte aligned); ncomp, nbox: [B] int32.  boxes: [B,max_boxes,8] float.
// hist: [B, 401] int32 and page_angle: [B] float workspaces (needed when max_angle > 0).
extern "C" int dlamd_db_postprocess(const void* mask, const void* prob, void* parent, void* comp_id, void* stats,
                                    void* ncomp, void* nbox, const float* page_scale, const int* valid, float* boxes,
                                    int B, int H, int W, int maxc, int max_boxes, float box_thresh,
                                    float unclip_ratio, int min_size, float max_angle, void* hist, float* page_angle,
                                    hipStream_t stream) {
  if (B <= 0) return 0;
  if ((long long)H * W > 0x7fffffffLL || B > 65535) return -1;
  const int hw = H * W;
  {
    if (max_angle > 0.f && (hist == nullptr || page_angle == nullptr)) return -2;
    int n = B * maxc;
    int nb = (max(max(n, B), B * ANG_BINS) + 255) / 256;
    hipLaunchKernelGGL(stats_init_kernel, dim3(nb), dim3(256), 0, stream, (int*)stats, (int*)ncomp, (int*)nbox, n, B,
                       max_angle > 0.f ? (int*)hist : nullptr);
  }
  hipLaunchKernelGGL(ccl_init_local_kernel, dim3((hw + 255) / 256, B), dim3(256), 0, stream, (const uint8_t*)mask,
                     (int*)parent, hw, B, W);
  dim3 grid((W + 63) / 64, (H + 3) / 4, B);
  hipLaunchKernelGGL(ccl_merge_kernel, grid, dim3(256), 0, stream, (const uint8_t*)mask, (int*)parent, H, W);
  hipLaunchKernelGGL(ccl_assign_kernel, grid, dim3(256), 0, stream, (int*)parent, (int*)comp_id, (int*)ncomp, H, W,
                     maxc);
  dim3 grid_s((W + 63) / 64, (H + STAT_ROWS - 1) / STAT_ROWS, B);
  hipLaunchKernelGGL(ccl_stats_kernel, grid_s, dim3(256), 0, stream, (const bf16_t*)prob, (const int*)parent,
                     (const int*)comp_id, (int*)stats, H, W, maxc);
  if (max_angle > 0.f) {
    hipLaunchKernelGGL(angle_vote_kernel, dim3((maxc + 255) / 256, B), dim3(256), 0, stream, (const int*)stats,
                       (const int*)ncomp, (int*)hist, maxc, box_thresh, min_size, max_angle);
    hipLaunchKernelGGL(angle_mode_kernel, dim3(B), dim3(64), 0, stream, (const int*)hist, page_angle);
  }
  hipLaunchKernelGGL(box_finalize_kernel, dim3((maxc + 255) / 256, B), dim3(256), 0, stream, (const int*)stats,
                     (const int*)ncomp, page_scale, valid, boxes, (int*)nbox, maxc, max_boxes, box_thresh,
                     unclip_ratio, min_size, max_angle, max_angle > 0.f ? (const float*)page_angle : nullptr, H, W);
  return DLAMD_CHECK_LAUNCH();
}
