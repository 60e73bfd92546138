// Serial host emulation of the GPU's self-synchronising decode of restart-less baseline JPEGs
// (csrc/kernels/jpeg_decode.hip, kernels a-e): the same subsequence decoder (csrc/kernels/jpeg_core.h), the same
// round structure inside a 256-subsequence chunk, the same seam pass, prefix sums and DC reconstruction - so the
// algorithm can be checked against libjpeg on a machine without a GPU.  Not a production decoder (it does the
// GPU's redundant work on one core).
#pragma once
#include <cmath>
#include <vector>

#include "../kernels/jpeg_core.h"
#include "jpeg_scan.h"

namespace dljpeg {

struct SyncStats {
  int32_t n_sub;
  int32_t max_rounds;        // most rounds any chunk needed before all its lanes had merged
  int32_t max_seam_steps;    // longest walk of a seam lane into its chunk
  int32_t blocks_decoded;    // must equal total blocks of the scan
  int32_t seam_passes, pad;
  int64_t sub_decodes;       // subsequence decodes in the synchronisation passes (work amplification = this / n_sub)
};

inline void idct8x8_host(const int16_t* c, uint8_t* out, int pitch) {
  static float basis[8][8];
  static bool init = false;
  if (!init) {
    for (int x = 0; x < 8; ++x)
      for (int u = 0; u < 8; ++u) basis[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979f / 16.f);
    init = true;
  }
  float tmp[64];
  for (int y = 0; y < 8; ++y)
    for (int u = 0; u < 8; ++u) {
      float a = 0.f;
      for (int v = 0; v < 8; ++v) a += basis[y][v] * (float)c[v * 8 + u];
      tmp[y * 8 + u] = a;
    }
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      float a = 0.f;
      for (int u = 0; u < 8; ++u) a += basis[x][u] * tmp[y * 8 + u];
      int v = (int)std::lrintf(a + 128.f);
      out[y * pitch + x] = (uint8_t)std::max(0, std::min(255, v));
    }
}

// out: gray plane of mcus_y*vs0*8 rows x mcus_x*hs0*8 columns.  Returns 0, -1 when the stream is not a baseline
// restart-less JPEG this path takes, -2 when `cap` is too small.
inline int decode_sync_host(const uint8_t* data, size_t n, uint8_t* out, size_t cap, SyncStats* stats) {
  JpegInfo info;
  JpegTables tables;
  std::vector<uint32_t> segs;
  if (scan_jpeg(data, n, &info, &tables, &segs) != 0) return -1;
  if (info.supported != 2) return -1;
  const int Wb = info.mcus_x * info.hs[0], Hb = info.mcus_y * info.vs[0];
  if ((size_t)Wb * Hb * 64 > cap) return -2;
  const int yblocks = info.hs[0] * info.vs[0];
  const int nblk = info.ncomp == 1 ? 1 : yblocks + 2;
  const int total_mcus = info.mcus_x * info.mcus_y;
  const int n_sub = (int)((info.scan_len + JPEG_SUB_BYTES - 1) / JPEG_SUB_BYTES);
  std::vector<int16_t> coef((size_t)Wb * Hb * 64, 0);
  std::vector<int32_t> dcdiff((size_t)total_mcus * yblocks, 0), sub_count(n_sub, 0), sub_off(n_sub, 0);
  std::vector<uint32_t> sub_state(n_sub, 0);
  ::JpegImageDesc d;
  memset(&d, 0, sizeof(d));
  d.data = data;
  d.tables = reinterpret_cast<const ::JpegTables*>(&tables);
  d.coef = coef.data();
  d.out = out;
  d.mcus_x = info.mcus_x;
  d.total_mcus = total_mcus;
  d.ncomp = info.ncomp;
  d.hs0 = info.hs[0];
  d.vs0 = info.vs[0];
  d.scan_end = (int32_t)(info.scan_off + info.scan_len);
  d.Wb = Wb;
  d.Hb = Hb;
  d.dcdiff = dcdiff.data();
  d.sub_state = sub_state.data();
  d.sub_count = sub_count.data();
  d.sub_off = sub_off.data();
  d.n_sub = n_sub;
  d.scan_off = (int32_t)info.scan_off;
  d.total_blocks = total_mcus * nblk;
  const uint16_t(*lut)[JPEG_LUT_SIZE] = d.tables->lut;
  const uint16_t* qt = d.tables->qt[0];
  SyncStats st_out = {n_sub, 0, 0, 0, 0, 0, 0};

  // a. per chunk: every lane decodes its own subsequence, then walks forward in lock-step rounds
  for (int j0 = 0; j0 < n_sub; j0 += JPEG_SUB_CHUNK) {
    const int n_here = std::min(JPEG_SUB_CHUNK, n_sub - j0);
    std::vector<SubState> st(n_here, SubState{0, 0, 0});
    std::vector<uint32_t> s_state(n_here);
    std::vector<int> s_cnt(n_here), s_stuff(n_here);
    std::vector<char> active(n_here, 1);
    for (int t = 0; t < n_here; ++t) s_stuff[t] = sub_stuffing(d, j0 + t);
    for (int t = 0; t < n_here; ++t) {
      s_cnt[t] = sub_decode<false>(d, j0 + t, st[t], s_stuff[t], 0, lut, qt, kZigzag);
      s_state[t] = sub_pack(st[t]);
      ++st_out.sub_decodes;
    }
    for (int r = 1; r < JPEG_SUB_CHUNK; ++r) {
      std::vector<uint32_t> mine(n_here);
      std::vector<int> cnt(n_here);
      std::vector<char> go(n_here, 0), same(n_here, 0);
      bool any = false;
      for (int t0 = 0; t0 < n_here; ++t0) {
        const int t = t0 + r;
        go[t0] = active[t0] && t < n_here;
        if (!go[t0]) continue;
        cnt[t0] = sub_decode<false>(d, j0 + t, st[t0], s_stuff[t], 0, lut, qt, kZigzag);
        mine[t0] = sub_pack(st[t0]);
        same[t0] = s_state[t] == mine[t0];
        ++st_out.sub_decodes;
      }
      for (int t0 = 0; t0 < n_here; ++t0) {
        if (go[t0]) { s_state[t0 + r] = mine[t0]; s_cnt[t0 + r] = cnt[t0]; }
        active[t0] = go[t0] && !same[t0];
        any = any || active[t0];
      }
      st_out.max_rounds = std::max(st_out.max_rounds, r);
      if (!any) break;
    }
    for (int t = 0; t < n_here; ++t) { sub_state[j0 + t] = s_state[t]; sub_count[j0 + t] = s_cnt[t]; }
  }
  // b. seams: passes until one changes nothing; all seam lanes of a pass start from a snapshot, one of the orders
  //    the concurrently running GPU lanes can take
  const int n_chunks = (n_sub + JPEG_SUB_CHUNK - 1) / JPEG_SUB_CHUNK;
  for (int pass = 0; pass < n_chunks; ++pass) {
    std::vector<uint32_t> snap = sub_state;
    bool changed = false;
    for (int c = 1; c < n_chunks; ++c) {
      const int j0 = c * JPEG_SUB_CHUNK, n_here = std::min(JPEG_SUB_CHUNK, n_sub - j0);
      SubState st = sub_unpack(snap[j0 - 1]);
      for (int t = 0; t < n_here; ++t) {
        const int cnt = sub_decode<false>(d, j0 + t, st, sub_stuffing(d, j0 + t), 0, lut, qt, kZigzag);
        const uint32_t mine = sub_pack(st);
        ++st_out.sub_decodes;
        sub_count[j0 + t] = cnt;
        if (sub_state[j0 + t] == mine) break;
        sub_state[j0 + t] = mine;
        changed = true;
        st_out.max_seam_steps = std::max(st_out.max_seam_steps, t + 1);
      }
    }
    st_out.seam_passes = pass + 1;
    if (!changed) break;
  }
  // c. where each subsequence writes
  int run = 0;
  for (int j = 0; j < n_sub; ++j) { sub_off[j] = run; run += sub_count[j]; }
  st_out.blocks_decoded = run;
  // d. coefficients
  for (int j = 0; j < n_sub; ++j) {
    SubState st = j == 0 ? SubState{0, 0, 0} : sub_unpack(sub_state[j - 1]);
    sub_decode<true>(d, j, st, sub_stuffing(d, j), sub_off[j], lut, qt, kZigzag);
  }
  // e. DC prefix sum, IDCT
  int dc = 0;
  for (int o = 0; o < total_mcus * yblocks; ++o) {
    dc += dcdiff[o];
    const int mcu = o / yblocks, b = o - mcu * yblocks;
    const int my = mcu / info.mcus_x, mx = mcu - my * info.mcus_x;
    const int bx = mx * d.hs0 + (b % d.hs0), by = my * d.vs0 + (b / d.hs0);
    const int v = dc * (int)qt[0];
    coef[((size_t)by * Wb + bx) * 64] = (int16_t)std::max(-32768, std::min(32767, v));
  }
  const int pitch = Wb * 8;
  for (int by = 0; by < Hb; ++by)
    for (int bx = 0; bx < Wb; ++bx) idct8x8_host(&coef[((size_t)by * Wb + bx) * 64], out + (size_t)by * 8 * pitch + bx * 8, pitch);
  if (stats) *stats = st_out;
  return 0;
}

}  // namespace dljpeg
