// DEFLATE (RFC 1951) / zlib (RFC 1950) decoder for page images.
//
// A 300-DPI A4 scan stored with FlateDecode is 8.7 MB (gray) or 26 MB (RGB) of raster per page; inflating it with
// zlib costs 20-60 ms of one core per page, which made lossless scans the slowest input of the whole service.  This
// decoder does the same job in less than half the time (8.7 MB gray page deflated at level 1, one 2 GHz Xeon core:
// 21-22 ms against 50 ms; level 6: 10-14 ms; ~29 cycles per symbol against a dependency chain of ~17):
//   * 64-bit bit buffer refilled with one unaligned 8-byte load (never per byte inside the main loop);
//   * one table lookup per symbol: 11-bit primary table for literals / lengths (entries carry the extra-bit count and
//     base, so a length needs no second table), 8-bit primary for distances, second-level tables for longer codes;
//   * a fast loop that runs while at least 32 input and 320 output bytes are left, so it needs no bounds checks per
//     symbol; the next symbol's table entry is loaded before the current match is copied; a length or distance leaves
//     the bit buffer together with its extra bits in one shift; matches are copied in 8-byte steps;
//   * it writes straight into the caller's buffer (the pinned staging memory of the page), no intermediate string.
// Anything it does not like (bad header, invalid code, output larger than the buffer) makes it return an error and the
// caller falls back to zlib, which keeps the tolerant behaviour for damaged files.  The Adler-32 trailer is not
// checked (the fallback path tolerates bad checksums as well).
//
// The reference gets its decoder from PyMuPDF (services/ocr_service/requirements.txt:214); nothing of it is in-tree.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace fastinflate {

constexpr int LIT_BITS = 11;     // primary table width, literal / length codes
constexpr int DIST_BITS = 8;     // primary table width, distance codes
constexpr int MAX_CODE_LEN = 15;

// Table entry (32 bits):
//   bits  0..7   bits consumed by this lookup: the code length, for a length / distance base PLUS its extra bits (one
//                shift drops both), for a subtable pointer the primary table width
//   bits  8..11  kind: 0 literal, 1 length / distance base, 2 end of block, 3 subtable pointer, 4 invalid
//   bits 12..15  kind 1: the code length alone (the extra bits are what lies above it in the consumed field);
//                kind 3: width of the subtable
//   bits 16..31  literal value | base value | subtable offset
struct Tables {
  uint32_t lit[(1 << LIT_BITS) + 2 * 288 * 2];   // primary + room for every possible subtable
  uint32_t dist[(1 << DIST_BITS) + 2 * 32 * 4];
};

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t v, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}

inline uint32_t make_entry(int sym, int len, bool is_dist) {
  if (is_dist) {
    if (sym >= 30) return 4u << 8 | (uint32_t)len;
    return (uint32_t)kDistBase[sym] << 16 | (1u | (uint32_t)len << 4) << 8 | (uint32_t)(len + kDistExtra[sym]);
  }
  if (sym < 256) return (uint32_t)sym << 16 | 0u << 8 | (uint32_t)len;
  if (sym == 256) return 2u << 8 | (uint32_t)len;
  if (sym >= 286) return 4u << 8 | (uint32_t)len;
  return (uint32_t)kLenBase[sym - 257] << 16 | (1u | (uint32_t)len << 4) << 8 | (uint32_t)(len + kLenExtra[sym - 257]);
}

// Canonical Huffman decoding table from code lengths.  Returns false for an over-subscribed code; an incomplete code
// is accepted (unused slots decode as invalid), as zlib does for the single-code distance case.
inline bool build_table(const uint8_t* lens, int n, uint32_t* table, int primary_bits, int table_cap, bool is_dist) {
  int count[MAX_CODE_LEN + 1] = {0};
  for (int i = 0; i < n; ++i) count[lens[i]]++;
  count[0] = 0;
  int left = 1;
  for (int l = 1; l <= MAX_CODE_LEN; ++l) {
    left = (left << 1) - count[l];
    if (left < 0) return false;
  }
  uint32_t next_code[MAX_CODE_LEN + 2];
  uint32_t code = 0;
  for (int l = 1; l <= MAX_CODE_LEN; ++l) {
    code = (code + count[l - 1]) << 1;
    next_code[l] = code;
  }
  const int psize = 1 << primary_bits;
  for (int i = 0; i < psize; ++i) table[i] = 4u << 8 | 1u;          // invalid, consumes one bit
  // short codes: replicate over the primary table
  int sub_next = psize;
  // first pass: codes that fit the primary table
  // second pass: longer codes, grouped by their primary_bits-bit prefix into subtables
  struct Long { uint32_t rev; uint8_t len; uint16_t sym; };
  std::vector<Long> longs;
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t c = next_code[l]++;
    const uint32_t rev = reverse_bits(c, l);
    if (l <= primary_bits) {
      const uint32_t e = make_entry(s, l, is_dist);
      for (uint32_t k = rev; k < (uint32_t)psize; k += 1u << l) table[k] = e;
    } else {
      longs.push_back({rev, (uint8_t)l, (uint16_t)s});
    }
  }
  if (longs.empty()) return true;
  // subtable width per prefix = longest code with that prefix - primary_bits
  std::vector<uint8_t> width(psize, 0);
  for (const Long& L : longs) {
    const uint32_t pre = L.rev & (psize - 1);
    const uint8_t w = (uint8_t)(L.len - primary_bits);
    if (w > width[pre]) width[pre] = w;
  }
  std::vector<int> base(psize, -1);
  for (const Long& L : longs) {
    const uint32_t pre = L.rev & (psize - 1);
    if (base[pre] < 0) {
      const int w = width[pre];
      if (sub_next + (1 << w) > table_cap) return false;
      base[pre] = sub_next;
      for (int k = 0; k < (1 << w); ++k) table[sub_next + k] = 4u << 8 | 1u;
      table[pre] = (uint32_t)sub_next << 16 | (3u | (uint32_t)w << 4) << 8 | (uint32_t)primary_bits;
      sub_next += 1 << w;
    }
    const int w = width[pre];
    const int sl = L.len - primary_bits;                       // bits this code uses inside the subtable
    const uint32_t e = make_entry(L.sym, sl, is_dist);
    for (uint32_t k = L.rev >> primary_bits; k < (1u << w); k += 1u << sl) table[base[pre] + k] = e;
  }
  return true;
}

struct BitReader {
  const uint8_t* in;
  const uint8_t* end;
  uint64_t buf = 0;
  int cnt = 0;
  int overrun = 0;          // zero bytes supplied past the end of the input
  inline void refill_fast() {          // needs in + 8 <= end
    uint64_t w;
    memcpy(&w, in, 8);
    buf |= w << cnt;
    in += (63 - cnt) >> 3;
    cnt |= 56;
  }
  inline void refill_safe() {
    while (cnt <= 56) {
      if (in < end) buf |= (uint64_t)*in++ << cnt;
      else ++overrun;
      cnt += 8;
    }
  }
  inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
  inline void drop(int n) { buf >>= n; cnt -= n; }
};

// Inflate a raw DEFLATE stream into dst[0 .. cap).  Returns bytes written, or -1 on a format error / overflow / input
// that ends before the final block does.
inline long long inflate_raw(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  BitReader br{src, src + n};
  uint8_t* out = dst;
  uint8_t* const out_end = dst + cap;
  std::vector<Tables> tabs(1);
  Tables& T = tabs[0];
  bool fixed_built = false;
  std::vector<Tables> fixed_tabs;
  bool last = false;
  while (!last) {
    br.refill_safe();
    last = br.peek(1);
    const int type = (br.peek(3) >> 1);
    br.drop(3);
    if (type == 0) {
      // stored block: byte-align, LEN / NLEN
      br.drop(br.cnt & 7);
      br.refill_safe();
      const uint32_t len = br.peek(16);
      br.drop(16);
      const uint32_t nlen = br.peek(16);
      br.drop(16);
      if ((len ^ 0xffffu) != nlen || br.overrun) return -1;
      // bytes still in the bit buffer belong to the block
      uint32_t left = len;
      while (left && br.cnt >= 8) {
        if (out >= out_end) return -1;
        *out++ = (uint8_t)br.peek(8);
        br.drop(8);
        --left;
      }
      if (left) {
        if ((size_t)(br.end - br.in) < left || (size_t)(out_end - out) < left) return -1;
        memcpy(out, br.in, left);
        br.in += left;
        out += left;
      }
      continue;
    }
    if (type == 3) return -1;
    const Tables* tp = &T;
    if (type == 1) {
      if (!fixed_built) {
        fixed_tabs.resize(1);
        uint8_t l[288 + 32];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        for (int i = 0; i < 32; ++i) l[288 + i] = 5;
        if (!build_table(l, 288, fixed_tabs[0].lit, LIT_BITS, (int)(sizeof(fixed_tabs[0].lit) / 4), false) ||
            !build_table(l + 288, 32, fixed_tabs[0].dist, DIST_BITS, (int)(sizeof(fixed_tabs[0].dist) / 4), true))
          return -1;
        fixed_built = true;
      }
      tp = &fixed_tabs[0];
    } else {
      br.refill_safe();
      const int hlit = br.peek(5) + 257;
      br.drop(5);
      const int hdist = br.peek(5) + 1;
      br.drop(5);
      const int hclen = br.peek(4) + 4;
      br.drop(4);
      if (hlit > 286 || hdist > 30) return -1;
      uint8_t cl[19] = {0};
      for (int i = 0; i < hclen; ++i) {
        br.refill_safe();
        cl[kOrder[i]] = (uint8_t)br.peek(3);
        br.drop(3);
      }
      uint32_t cltab[1 << 7];
      if (!build_table(cl, 19, cltab, 7, 1 << 7, false)) return -1;
      uint8_t lens[288 + 32] = {0};
      int i = 0;
      while (i < hlit + hdist) {
        br.refill_safe();
        const uint32_t e = cltab[br.peek(7)];
        const int kind = (e >> 8) & 0xf;
        if (kind != 0) return -1;                    // code-length symbols are 0..18: all "literals" here
        br.drop(e & 0xff);
        const int sym = e >> 16;
        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) {
          if (i == 0) return -1;
          val = lens[i - 1];
          rep = 3 + br.peek(2);
          br.drop(2);
        } else if (sym == 17) {
          rep = 3 + br.peek(3);
          br.drop(3);
        } else {
          rep = 11 + br.peek(7);
          br.drop(7);
        }
        if (i + rep > hlit + hdist) return -1;
        while (rep--) lens[i++] = (uint8_t)val;
      }
      if (br.overrun) return -1;
      if (lens[256] == 0) return -1;
      if (!build_table(lens, hlit, T.lit, LIT_BITS, (int)(sizeof(T.lit) / 4), false) ||
          !build_table(lens + hlit, hdist, T.dist, DIST_BITS, (int)(sizeof(T.dist) / 4), true))
        return -1;
    }
    const uint32_t* lit = tp->lit;
    const uint32_t* dtab = tp->dist;
    // ---- block body, fast part: at least 32 input and 320 output bytes left, so no per-symbol bounds checks (an
    // iteration refills at most three times, 7 bytes each, and writes at most 258 + 16 bytes).  The entry of the NEXT
    // symbol is looked up before the current match is copied, so its load overlaps the copy; a length or distance and
    // its extra bits leave the buffer in one shift (the extra bits are cut out of the saved buffer).  Between two
    // refills at most 48 bits go: a length via a subtable (11 + 4 + 5) and a distance via one (8 + 7 + 13).
    bool block_done = false;
    constexpr uint64_t LMASK = (1u << LIT_BITS) - 1, DMASK = (1u << DIST_BITS) - 1;
    while (!block_done) {
    if ((br.end - br.in) >= 32 && (out_end - out) >= 320) {
      br.refill_fast();
      uint32_t e = lit[br.buf & LMASK];
      do {
        uint64_t saved = br.buf;
        uint32_t kind = (e >> 8) & 0xf;
        br.drop(e & 0xff);
        if (kind == 0) {
          *out++ = (uint8_t)(e >> 16);
          e = lit[br.buf & LMASK];
          if (((e >> 8) & 0xf) == 0) {
            br.drop(e & 0xff);
            *out++ = (uint8_t)(e >> 16);
            e = lit[br.buf & LMASK];
            if (((e >> 8) & 0xf) == 0) {
              br.drop(e & 0xff);
              *out++ = (uint8_t)(e >> 16);
              br.refill_fast();
              e = lit[br.buf & LMASK];
              continue;
            }
          }
          br.refill_fast();                        // e was looked up from bits that are still at the bottom of the buffer
          continue;
        }
        if (kind == 3) {
          e = lit[(e >> 16) + (uint32_t)(br.buf & ((1u << ((e >> 12) & 0xf)) - 1))];
          saved = br.buf;
          kind = (e >> 8) & 0xf;
          br.drop(e & 0xff);
          if (kind == 0) {
            *out++ = (uint8_t)(e >> 16);
            br.refill_fast();
            e = lit[br.buf & LMASK];
            continue;
          }
        }
        if (kind == 2) {
          block_done = true;
          break;
        }
        if (kind != 1) return -1;
        const uint32_t len = (e >> 16) + (uint32_t)((saved & ((1ull << (e & 0xff)) - 1)) >> ((e >> 12) & 0xf));
        uint32_t d = dtab[br.buf & DMASK];
        uint32_t dk = (d >> 8) & 0xf;
        if (dk == 3) {
          br.drop(DIST_BITS);
          d = dtab[(d >> 16) + (uint32_t)(br.buf & ((1u << ((d >> 12) & 0xf)) - 1))];
          dk = (d >> 8) & 0xf;
        }
        saved = br.buf;
        br.drop(d & 0xff);
        if (dk != 1) return -1;
        const uint32_t dist = (d >> 16) + (uint32_t)((saved & ((1ull << (d & 0xff)) - 1)) >> ((d >> 12) & 0xf));
        if (dist > (size_t)(out - dst)) return -1;
        br.refill_fast();
        e = lit[br.buf & LMASK];                   // the next symbol's entry is on its way while the match is copied
        const uint8_t* from = out - dist;
        uint8_t* o = out;
        out += len;
        if (dist >= 8) {
          // whole 8-byte steps may run past the match end (out_end - out >= 320 > 258 + 16); a step reads what an
          // earlier step wrote when dist < len, which is what an overlapping match means
          uint64_t w;
          memcpy(&w, from, 8);
          memcpy(o, &w, 8);
          memcpy(&w, from + 8, 8);
          memcpy(o + 8, &w, 8);
          if (len > 16) {
            from += 16;
            o += 16;
            do {
              memcpy(&w, from, 8);
              memcpy(o, &w, 8);
              from += 8;
              o += 8;
            } while (o < out);
          }
        } else if (dist == 1) {
          memset(o, *from, len);
        } else {
          do { *o++ = *from++; } while (o < out);
        }
      } while ((br.end - br.in) >= 32 && (out_end - out) >= 320);
    }
    if (block_done) break;
    // ---- block body, careful part (one symbol, then the fast part is tried again): the last bytes of the input or
    // of the output buffer
    {
      br.refill_safe();
      if (br.overrun > 8) return -1;              // decoding bytes that are not there: a truncated stream
      uint32_t e = lit[br.peek(LIT_BITS)];
      uint32_t kind = (e >> 8) & 0xf;
      if (kind == 3) {
        br.drop(LIT_BITS);
        e = lit[(e >> 16) + br.peek((e >> 12) & 0xf)];
        kind = (e >> 8) & 0xf;
      }
      if (kind == 0) {
        br.drop(e & 0xff);
        if (out >= out_end) return -1;
        *out++ = (uint8_t)(e >> 16);
        continue;
      }
      if (kind == 2) {
        br.drop(e & 0xff);
        block_done = true;
        break;
      }
      if (kind != 1) return -1;
      const uint32_t len = (e >> 16) + (br.peek(e & 0xff) >> ((e >> 12) & 0xf));
      br.drop(e & 0xff);
      br.refill_safe();
      uint32_t d = dtab[br.peek(DIST_BITS)];
      uint32_t dk = (d >> 8) & 0xf;
      if (dk == 3) {
        br.drop(DIST_BITS);
        d = dtab[(d >> 16) + br.peek((d >> 12) & 0xf)];
        dk = (d >> 8) & 0xf;
      }
      if (dk != 1) return -1;
      const uint32_t dist = (d >> 16) + (br.peek(d & 0xff) >> ((d >> 12) & 0xf));
      br.drop(d & 0xff);
      if (dist > (size_t)(out - dst) || (size_t)(out_end - out) < len) return -1;
      const uint8_t* from = out - dist;
      for (uint32_t k = 0; k < len; ++k) out[k] = from[k];
      out += len;
    }
    }
    if (br.overrun > 8) return -1;                // the block ended inside bytes that are not there
  }
  if (br.overrun > 8) return -1;
  return (long long)(out - dst);
}

// zlib wrapper: 2-byte header (CM = 8, FCHECK), no preset dictionary.  The trailer is ignored.
inline long long inflate_zlib(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  if (n < 2) return -1;
  const unsigned cmf = src[0], flg = src[1];
  if ((cmf & 0x0f) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return -1;
  return inflate_raw(src + 2, n - 2, dst, cap);
}

}  // namespace fastinflate
