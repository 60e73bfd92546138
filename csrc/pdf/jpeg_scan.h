// JPEG bitstream header parser: everything the GPU decoder (csrc/kernels/jpeg_decode.hip)
// needs to decode a baseline JPEG in parallel - frame geometry, quantisation tables,
// Huffman lookup tables, and the byte offset of every restart segment.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace dljpeg {

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

constexpr int LUT_BITS = 10;
constexpr int LUT_SIZE = 1 << LUT_BITS;

// Plain-old-data header shared with Python (ctypes) and the HIP kernel.  Keep layout in sync
// with ops/jpeg.py::JPEG_INFO_DTYPE and jpeg_decode.hip::JpegTables.
struct JpegInfo {
  int32_t width, height;
  int32_t ncomp;             // 1 (gray) or 3 (YCbCr)
  int32_t progressive;       // 1: SOF2 (not decodable on the GPU path)
  int32_t precision;         // bits per sample (8 supported)
  int32_t restart_interval;  // MCUs per restart segment (0: none)
  int32_t mcu_w, mcu_h;      // MCU size in pixels
  int32_t mcus_x, mcus_y;
  int32_t hs[4], vs[4];      // sampling factors per component
  int32_t tq[4], td[4], ta[4];
  int32_t blocks_per_mcu;
  int32_t n_segments;
  int64_t scan_off, scan_len;  // entropy-coded data inside the stream
  int32_t supported;           // GPU decoder: 0 no, 1 one lane per restart segment, 2 self-synchronising (no restarts)
  int32_t pad;
};

// Device-side tables for one image.
//   qt[c][k]      quantiser for component c, natural (row-major) order
//   lut[t][code]  t = 0 DC-luma, 1 AC-luma, 2 DC-chroma, 3 AC-chroma: (len << 8) | symbol for
//                 codes of <= LUT_BITS bits, 0 when longer
//   maxcode/valptr/mincode + vals: canonical tables for the slow path (codes of 11..16 bits)
struct JpegTables {
  uint16_t qt[3][64];
  uint16_t lut[4][LUT_SIZE];
  int32_t maxcode[4][18];
  int32_t valoff[4][17];
  uint8_t vals[4][256];
};

struct Huff {
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  bool present = false;
};

inline void build_lut(const Huff& h, uint16_t* lut, int32_t* maxcode, int32_t* valoff, uint8_t* vals) {
  memset(lut, 0, sizeof(uint16_t) * LUT_SIZE);
  memcpy(vals, h.vals, 256);
  int code = 0, k = 0;
  for (int len = 1; len <= 16; ++len) {
    valoff[len] = k - code;
    for (int i = 0; i < h.bits[len]; ++i, ++k, ++code) {
      if (len <= LUT_BITS) {
        int shift = LUT_BITS - len;
        if (code >= (1 << len)) return;  // over-subscribed table (rejected by the parser; defence in depth)
        for (int f = 0; f < (1 << shift); ++f) lut[(code << shift) | f] = (uint16_t)((len << 8) | h.vals[k]);
      }
    }
    maxcode[len] = h.bits[len] ? code - 1 : -1;
    code <<= 1;
  }
  maxcode[17] = 0x7fffffff;
  maxcode[0] = -1;
  valoff[0] = 0;
}

// Returns 0 on success.  seg_offsets receives the start of every restart segment
// (relative to the start of ``d``); the entropy data ends at scan_off + scan_len.
inline int scan_jpeg(const uint8_t* d, size_t n, JpegInfo* info, JpegTables* tables, std::vector<uint32_t>* seg_offsets) {
  memset(info, 0, sizeof(*info));
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return -1;
  uint16_t qt[4][64];
  bool qt_ok[4] = {false, false, false, false};
  Huff huff[2][4];
  size_t p = 2;
  bool have_sof = false;
  int comp_id[4] = {0, 0, 0, 0};
  while (p + 4 <= n) {
    if (d[p] != 0xFF) { ++p; continue; }
    uint8_t m = d[p + 1];
    if (m == 0xFF) { ++p; continue; }
    p += 2;
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    if (m == 0xD9) break;
    if (p + 2 > n) return -2;
    size_t len = ((size_t)d[p] << 8) | d[p + 1];
    if (len < 2 || p + len > n) return -2;
    const uint8_t* s = d + p + 2;
    size_t sl = len - 2;
    if (m == 0xDB) {
      size_t q = 0;
      while (q < sl) {
        int pq = s[q] >> 4, tq = s[q] & 15;
        ++q;
        if (tq > 3) return -3;
        for (int k = 0; k < 64; ++k) {
          if (q + (pq ? 2 : 1) > sl) return -3;
          uint16_t v = pq ? (uint16_t)((s[q] << 8) | s[q + 1]) : s[q];
          q += pq ? 2 : 1;
          qt[tq][kZigzag[k]] = v;
        }
        qt_ok[tq] = true;
      }
    } else if (m == 0xC4) {
      size_t q = 0;
      while (q + 17 <= sl) {
        int tc = s[q] >> 4, th = s[q] & 15;
        if (tc > 1 || th > 3) return -4;
        Huff& h = huff[tc][th];
        int total = 0;
        // Kraft check: a length-k prefix code has at most 2^k codewords and every codeword of a
        // shorter length removes two of the next length.  An over-subscribed table would make
        // build_lut() index past its 2^LUT_BITS entries.
        long long code = 0;
        for (int k = 1; k <= 16; ++k) {
          h.bits[k] = s[q + k];
          total += s[q + k];
          code += s[q + k];
          if (code > (1LL << k)) return -4;
          code <<= 1;
        }
        q += 17;
        if (total > 256 || q + total > sl) return -4;
        memcpy(h.vals, s + q, total);
        q += total;
        h.present = true;
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      if (sl < 6) return -5;
      info->progressive = (m == 0xC2);
      info->precision = s[0];
      info->height = (s[1] << 8) | s[2];
      info->width = (s[3] << 8) | s[4];
      info->ncomp = s[5];
      if (info->ncomp != 1 && info->ncomp != 3) return -5;
      if (sl < 6 + 3 * (size_t)info->ncomp) return -5;
      for (int c = 0; c < info->ncomp; ++c) {
        comp_id[c] = s[6 + 3 * c];
        info->hs[c] = s[7 + 3 * c] >> 4;
        info->vs[c] = s[7 + 3 * c] & 15;
        info->tq[c] = s[8 + 3 * c] & 3;
        if (info->hs[c] < 1 || info->hs[c] > 2 || info->vs[c] < 1 || info->vs[c] > 2) return -5;
      }
      have_sof = true;
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return -6;  // lossless / arithmetic / hierarchical
    } else if (m == 0xDD) {
      if (sl < 2) return -7;
      info->restart_interval = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (!have_sof || sl < 1) return -8;
      int ns = s[0];
      if (sl < 1 + 2 * (size_t)ns + 3) return -8;
      if (ns < 1 || ns > info->ncomp) return -8;
      for (int k = 0; k < ns; ++k) {
        int cid = s[1 + 2 * k];
        int td = s[2 + 2 * k] >> 4, ta = s[2 + 2 * k] & 15;
        if (td > 3 || ta > 3) return -8;  // selectors index huff[2][4]
        bool matched = false;
        for (int c = 0; c < info->ncomp; ++c)
          if (comp_id[c] == cid) { info->td[c] = td; info->ta[c] = ta; matched = true; }
        if (!matched) return -8;  // a scan component that the frame header does not declare
      }
      info->scan_off = (int64_t)(p + len);
      // one interleaved scan with all components = baseline layout
      bool all_in_scan = (ns == info->ncomp);
      // geometry
      int hmax = 1, vmax = 1;
      if (info->ncomp == 1) { info->hs[0] = info->vs[0] = 1; }
      for (int c = 0; c < info->ncomp; ++c) { hmax = std::max(hmax, info->hs[c]); vmax = std::max(vmax, info->vs[c]); }
      info->mcu_w = 8 * hmax;
      info->mcu_h = 8 * vmax;
      info->mcus_x = (info->width + info->mcu_w - 1) / info->mcu_w;
      info->mcus_y = (info->height + info->mcu_h - 1) / info->mcu_h;
      info->blocks_per_mcu = 0;
      for (int c = 0; c < info->ncomp; ++c) info->blocks_per_mcu += info->hs[c] * info->vs[c];
      // chroma must be 1x1 so that component order inside the MCU is Y.., Cb, Cr
      bool chroma_ok = info->ncomp == 1 || (info->hs[1] == 1 && info->vs[1] == 1 && info->hs[2] == 1 && info->vs[2] == 1);
      // restart segments
      size_t q = (size_t)info->scan_off;
      if (seg_offsets) { seg_offsets->clear(); seg_offsets->push_back((uint32_t)q); }
      size_t end = n;
      while (q + 1 < n) {
        const uint8_t* f = (const uint8_t*)memchr(d + q, 0xFF, n - q - 1);
        if (!f) { end = n; break; }
        q = (size_t)(f - d);
        uint8_t mk = d[q + 1];
        if (mk == 0x00 || mk == 0xFF) { q += (mk == 0 ? 2 : 1); continue; }
        if (mk >= 0xD0 && mk <= 0xD7) {
          if (seg_offsets) seg_offsets->push_back((uint32_t)(q + 2));
          q += 2;
          continue;
        }
        end = q;  // EOI or any other marker terminates the scan
        break;
      }
      info->scan_len = (int64_t)end - info->scan_off;
      info->n_segments = seg_offsets ? (int32_t)seg_offsets->size() : 0;
      bool tables_ok = true;
      for (int c = 0; c < info->ncomp; ++c)
        tables_ok = tables_ok && qt_ok[info->tq[c]] && huff[0][info->td[c]].present && huff[1][info->ta[c]].present;
      long long total_mcus = (long long)info->mcus_x * info->mcus_y;
      bool seg_ok = true;
      if (info->restart_interval > 0) {
        long long expect = (total_mcus + info->restart_interval - 1) / info->restart_interval;
        seg_ok = (long long)info->n_segments == expect;
      } else {
        seg_ok = info->n_segments == 1;
      }
      const bool base_ok = !info->progressive && info->precision == 8 && all_in_scan && chroma_ok && tables_ok && seg_ok &&
                           info->scan_len > 0 && (long long)info->scan_off + info->scan_len < (1LL << 31);
      // One lane per restart segment needs many segments to fill a GPU; a restart-less scan goes through the
      // self-synchronising decoder; a large image cut into a handful of long segments stays on the host.
      info->supported = 0;
      if (base_ok) {
        if (info->restart_interval == 0) info->supported = 2;
        else if (info->n_segments >= 32 || total_mcus <= 4096) info->supported = 1;
      }
      if (tables && tables_ok) {
        memset(tables, 0, sizeof(*tables));
        for (int c = 0; c < info->ncomp && c < 3; ++c) memcpy(tables->qt[c], qt[info->tq[c]], sizeof(uint16_t) * 64);
        // table slots: 0/1 = DC/AC of component 0, 2/3 = DC/AC of component 1 (chroma; comp 2 must share)
        build_lut(huff[0][info->td[0]], tables->lut[0], tables->maxcode[0], tables->valoff[0], tables->vals[0]);
        build_lut(huff[1][info->ta[0]], tables->lut[1], tables->maxcode[1], tables->valoff[1], tables->vals[1]);
        if (info->ncomp == 3) {
          if (info->td[2] != info->td[1] || info->ta[2] != info->ta[1]) info->supported = 0;
          build_lut(huff[0][info->td[1]], tables->lut[2], tables->maxcode[2], tables->valoff[2], tables->vals[2]);
          build_lut(huff[1][info->ta[1]], tables->lut[3], tables->maxcode[3], tables->valoff[3], tables->vals[3]);
        }
      }
      return 0;
    }
    p += len;
  }
  return -9;  // no SOS
}

}  // namespace dljpeg
