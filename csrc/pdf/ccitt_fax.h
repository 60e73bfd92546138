// CCITTFaxDecode (ITU-T T.4 / T.6): the bilevel codec most office scanners put into PDF files.
//
// Role in the reference stack: PyMuPDF decodes these images behind PymuDocDataset
// (services/ocr_service/ocr_service.py:210); here it is one more filter of the native reader.
// Supported: K < 0 (Group 4, pure two-dimensional), K = 0 (Group 3 one-dimensional), K > 0 (Group 3 mixed, EOL +
// tag bit), /Columns, /Rows, /BlackIs1, /EncodedByteAlign.  Output: packed 1-bit rows, MSB first, exactly what the
// filter hands to the image dictionary (bit = 1 is white unless BlackIs1).
// Malformed input never throws past the caller and never reads or writes out of bounds: a damaged line is
// completed with white and decoding stops at the first unrecoverable code (the remaining rows stay white).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace dlpdf {
namespace ccitt {

struct Code { const char* bits; int run; };

// terminating codes 0..63 followed by make-up codes 64..1728 (T.4 tables 2 and 3)
static const Code kWhite[] = {
    {"00110101", 0},  {"000111", 1},    {"0111", 2},      {"1000", 3},      {"1011", 4},      {"1100", 5},      {"1110", 6},
    {"1111", 7},      {"10011", 8},     {"10100", 9},     {"00111", 10},    {"01000", 11},    {"001000", 12},   {"000011", 13},
    {"110100", 14},   {"110101", 15},   {"101010", 16},   {"101011", 17},   {"0100111", 18},  {"0001100", 19},  {"0001000", 20},
    {"0010111", 21},  {"0000011", 22},  {"0000100", 23},  {"0101000", 24},  {"0101011", 25},  {"0010011", 26},  {"0100100", 27},
    {"0011000", 28},  {"00000010", 29}, {"00000011", 30}, {"00011010", 31}, {"00011011", 32}, {"00010010", 33}, {"00010011", 34},
    {"00010100", 35}, {"00010101", 36}, {"00010110", 37}, {"00010111", 38}, {"00101000", 39}, {"00101001", 40}, {"00101010", 41},
    {"00101011", 42}, {"00101100", 43}, {"00101101", 44}, {"00000100", 45}, {"00000101", 46}, {"00001010", 47}, {"00001011", 48},
    {"01010010", 49}, {"01010011", 50}, {"01010100", 51}, {"01010101", 52}, {"00100100", 53}, {"00100101", 54}, {"01011000", 55},
    {"01011001", 56}, {"01011010", 57}, {"01011011", 58}, {"01001010", 59}, {"01001011", 60}, {"00110010", 61}, {"00110011", 62},
    {"00110100", 63},
    {"11011", 64},       {"10010", 128},      {"010111", 192},     {"0110111", 256},    {"00110110", 320},   {"00110111", 384},
    {"01100100", 448},   {"01100101", 512},   {"01101000", 576},   {"01100111", 640},   {"011001100", 704},  {"011001101", 768},
    {"011010010", 832},  {"011010011", 896},  {"011010100", 960},  {"011010101", 1024}, {"011010110", 1088}, {"011010111", 1152},
    {"011011000", 1216}, {"011011001", 1280}, {"011011010", 1344}, {"011011011", 1408}, {"010011000", 1472}, {"010011001", 1536},
    {"010011010", 1600}, {"011000", 1664},    {"010011011", 1728},
};

static const Code kBlack[] = {
    {"0000110111", 0},    {"010", 1},           {"11", 2},            {"10", 3},            {"011", 4},           {"0011", 5},
    {"0010", 6},          {"00011", 7},         {"000101", 8},        {"000100", 9},        {"0000100", 10},      {"0000101", 11},
    {"0000111", 12},      {"00000100", 13},     {"00000111", 14},     {"000011000", 15},    {"0000010111", 16},   {"0000011000", 17},
    {"0000001000", 18},   {"00001100111", 19},  {"00001101000", 20},  {"00001101100", 21},  {"00000110111", 22},  {"00000101000", 23},
    {"00000010111", 24},  {"00000011000", 25},  {"000011001010", 26}, {"000011001011", 27}, {"000011001100", 28}, {"000011001101", 29},
    {"000001101000", 30}, {"000001101001", 31}, {"000001101010", 32}, {"000001101011", 33}, {"000011010010", 34}, {"000011010011", 35},
    {"000011010100", 36}, {"000011010101", 37}, {"000011010110", 38}, {"000011010111", 39}, {"000001101100", 40}, {"000001101101", 41},
    {"000011011010", 42}, {"000011011011", 43}, {"000001010100", 44}, {"000001010101", 45}, {"000001010110", 46}, {"000001010111", 47},
    {"000001100100", 48}, {"000001100101", 49}, {"000001010010", 50}, {"000001010011", 51}, {"000000100100", 52}, {"000000110111", 53},
    {"000000111000", 54}, {"000000100111", 55}, {"000000101000", 56}, {"000001011000", 57}, {"000001011001", 58}, {"000000101011", 59},
    {"000000101100", 60}, {"000001011010", 61}, {"000001100110", 62}, {"000001100111", 63},
    {"0000001111", 64},     {"000011001000", 128},   {"000011001001", 192},   {"000001011011", 256},   {"000000110011", 320},
    {"000000110100", 384},  {"000000110101", 448},   {"0000001101100", 512},  {"0000001101101", 576},  {"0000001001010", 640},
    {"0000001001011", 704}, {"0000001001100", 768},  {"0000001001101", 832},  {"0000001110010", 896},  {"0000001110011", 960},
    {"0000001110100", 1024}, {"0000001110101", 1088}, {"0000001110110", 1152}, {"0000001110111", 1216}, {"0000001010010", 1280},
    {"0000001010011", 1344}, {"0000001010100", 1408}, {"0000001010101", 1472}, {"0000001011010", 1536}, {"0000001011011", 1600},
    {"0000001100100", 1664}, {"0000001100101", 1728},
};

// extended make-up codes, common to both colours
static const Code kExtended[] = {
    {"00000001000", 1792},  {"00000001100", 1856},  {"00000001101", 1920},  {"000000010010", 1984}, {"000000010011", 2048},
    {"000000010100", 2112}, {"000000010101", 2176}, {"000000010110", 2240}, {"000000010111", 2304}, {"000000011100", 2368},
    {"000000011101", 2432}, {"000000011110", 2496}, {"000000011111", 2560},
};

constexpr int kLutBits = 13;

struct RunLut {
  // entry = (run << 4) | code length; 0 = no code with this prefix
  uint32_t e[1 << kLutBits];
};

inline void lut_add(RunLut& l, const Code& c) {
  const int len = (int)strlen(c.bits);
  uint32_t v = 0;
  for (int i = 0; i < len; ++i) v = (v << 1) | (uint32_t)(c.bits[i] - '0');
  const uint32_t lo = v << (kLutBits - len), n = 1u << (kLutBits - len);
  for (uint32_t k = 0; k < n; ++k) l.e[lo + k] = ((uint32_t)c.run << 4) | (uint32_t)len;
}

struct RunLuts {
  RunLut l[2];
  RunLuts() {
    memset(l, 0, sizeof l);
    for (const Code& c : kWhite) lut_add(l[0], c);
    for (const Code& c : kBlack) lut_add(l[1], c);
    for (const Code& c : kExtended) { lut_add(l[0], c); lut_add(l[1], c); }
  }
};

inline const RunLut& run_lut(int black) {
  static const RunLuts luts;      // thread-safe one-time construction (decode runs on a thread pool)
  return luts.l[black];
}

struct Bits {
  const uint8_t* p;
  size_t n, pos = 0;       // pos in bits
  Bits(const uint8_t* d, size_t len) : p(d), n(len) {}
  bool eof() const { return pos >= n * 8; }
  // next k bits (k <= 24) without consuming, zero padded past the end
  uint32_t peek(int k) const {
    const size_t byte = pos >> 3;
    if (byte + 8 <= n) {                 // one unaligned 8-byte load, big-endian: the next bit is the top bit
      uint64_t w;
      memcpy(&w, p + byte, 8);
      return (uint32_t)((__builtin_bswap64(w) << (pos & 7)) >> (64 - k));
    }
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) v = (v << 8) | (byte + i < n ? p[byte + i] : 0u);
    return (v << (pos & 7)) >> (32 - k);
  }
  void skip(int k) { pos += (size_t)k; }
  void align() { pos = (pos + 7) & ~(size_t)7; }
};

// One run of colour ``black`` (make-up codes summed with the terminating code); -1 on an invalid code.
inline int read_run(Bits& b, int black) {
  const RunLut& l = run_lut(black);
  int total = 0;
  for (int guard = 0; guard < 64; ++guard) {
    const uint32_t e = l.e[b.peek(kLutBits)];
    if (e == 0) return -1;
    b.skip((int)(e & 15));
    const int run = (int)(e >> 4);
    total += run;
    if (run < 64) return total;
  }
  return -1;
}

enum Mode { PASS, HORIZ, V0, VR1, VR2, VR3, VL1, VL2, VL3, EOL_OR_BAD };

inline Mode read_mode(Bits& b) {
  const uint32_t v = b.peek(7);
  if (v & 0x40) { b.skip(1); return V0; }                 // 1
  if ((v >> 4) == 3) { b.skip(3); return VR1; }           // 011
  if ((v >> 4) == 2) { b.skip(3); return VL1; }           // 010
  if ((v >> 4) == 1) { b.skip(3); return HORIZ; }         // 001
  if ((v >> 3) == 1) { b.skip(4); return PASS; }          // 0001
  if ((v >> 1) == 3) { b.skip(6); return VR2; }           // 000011
  if ((v >> 1) == 2) { b.skip(6); return VL2; }           // 000010
  if (v == 3) { b.skip(7); return VR3; }                  // 0000011
  if (v == 2) { b.skip(7); return VL3; }                  // 0000010
  return EOL_OR_BAD;
}

// Skips fill zeros and one EOL (000000000001) if present at the cursor. Returns true when an EOL was consumed.
inline bool skip_eol(Bits& b) {
  size_t save = b.pos;
  int zeros = 0;
  while (!b.eof() && b.peek(1) == 0) { b.skip(1); ++zeros; }
  if (!b.eof() && zeros >= 11) { b.skip(1); return true; }
  b.pos = save;
  return false;
}

// changes: positions where the colour flips, starting from white at column 0, strictly usable up to W; padded
// with W entries so that look-ups past the end are safe.
inline bool decode_1d_row(Bits& b, int W, std::vector<int>& cur) {
  cur.clear();
  int a0 = 0, black = 0;
  while (a0 < W) {
    const int run = read_run(b, black);
    if (run < 0) return false;
    a0 += run;
    if (a0 > W) a0 = W;
    cur.push_back(a0);
    black ^= 1;
  }
  return true;
}

inline bool decode_2d_row(Bits& b, int W, const std::vector<int>& ref, std::vector<int>& cur) {
  cur.clear();
  int a0 = -1, black = 0;
  size_t bi = 0;            // index into ref of b1: first change right of a0 whose parity matches ``black``
  const auto refat = [&](size_t i) { return i < ref.size() ? ref[i] : W; };
  while (a0 < W) {
    // b1: ref changes at even indices go white -> black; the change must be to the colour opposite to ours
    while (bi > 0 && refat(bi - 1) > a0) --bi;               // a vertical-left step may have moved a0 backwards
    while (refat(bi) <= a0 && refat(bi) < W) ++bi;
    if ((bi & 1) != (size_t)black) ++bi;                      // parity: even index = transition out of white
    // (after the parity fix the entry is still right of a0 because positions are non-decreasing)
    const int b1 = refat(bi), b2 = refat(bi + 1);
    const Mode m = read_mode(b);
    if (m == PASS) {
      a0 = b2;
      continue;
    }
    if (m == HORIZ) {
      const int r1 = read_run(b, black);
      if (r1 < 0) return false;
      const int r2 = read_run(b, black ^ 1);
      if (r2 < 0) return false;
      int s = a0 < 0 ? 0 : a0;
      int a1 = s + r1;
      if (a1 > W) a1 = W;
      int a2 = a1 + r2;
      if (a2 > W) a2 = W;
      cur.push_back(a1);
      cur.push_back(a2);
      a0 = a2;
      continue;
    }
    int d;
    switch (m) {
      case V0: d = 0; break;
      case VR1: d = 1; break;
      case VR2: d = 2; break;
      case VR3: d = 3; break;
      case VL1: d = -1; break;
      case VL2: d = -2; break;
      case VL3: d = -3; break;
      default: return false;
    }
    int a1 = b1 + d;
    if (a1 > W) a1 = W;
    const int floor_ = cur.empty() ? 0 : cur.back();
    if (a1 < floor_) a1 = floor_;        // damaged data: never let the change list run backwards
    cur.push_back(a1);
    a0 = a1;
    black ^= 1;
  }
  return true;
}

inline void paint_row(const std::vector<int>& changes, int W, bool black_is_1, uint8_t* row, size_t rowbytes) {
  // start from all white; white sample bit = black_is_1 ? 0 : 1
  memset(row, black_is_1 ? 0x00 : 0xFF, rowbytes);
  int x = 0, black = 0;
  for (size_t i = 0; i <= changes.size() && x < W; ++i) {
    const int nx = i < changes.size() ? (changes[i] < W ? changes[i] : W) : W;
    if (black && nx > x) {
      for (int k = x; k < nx;) {
        if ((k & 7) == 0 && nx - k >= 8) { row[k >> 3] = black_is_1 ? 0xFF : 0x00; k += 8; continue; }
        if (black_is_1) row[k >> 3] |= (uint8_t)(0x80 >> (k & 7));
        else row[k >> 3] &= (uint8_t)~(0x80 >> (k & 7));
        ++k;
      }
    }
    if (nx > x) x = nx;
    black ^= 1;
  }
}

// rows <= 0: decode until the data ends (bounded by max_rows).
inline std::string decode(const uint8_t* data, size_t n, int K, int columns, int rows, bool black_is_1, bool byte_align,
                          int max_rows = 1 << 16) {
  if (columns <= 0 || columns > (1 << 16)) return std::string();
  const size_t rowbytes = ((size_t)columns + 7) / 8;
  const int want = rows > 0 ? (rows < max_rows ? rows : max_rows) : max_rows;
  std::string out;
  out.reserve(rowbytes * (size_t)(rows > 0 ? want : 1024));
  Bits b(data, n);
  std::vector<int> ref, cur;
  ref.reserve(256);
  cur.reserve(256);
  int done = 0;
  bool ok = true;
  while (done < want && ok && !b.eof()) {
    if (byte_align) b.align();
    bool two_d = K < 0;
    if (K >= 0) {
      const bool eol = skip_eol(b);
      if (K > 0) {
        if (eol) { two_d = b.peek(1) == 0; b.skip(1); }
        else two_d = false;
      }
      if (eol && b.eof()) break;
    } else {
      // EOFB = two EOLs
      if (b.peek(12) == 1 && ((b.peek(24) & 0xFFF) == 1)) break;
    }
    ok = two_d ? decode_2d_row(b, columns, ref, cur) : decode_1d_row(b, columns, cur);
    if (!ok && cur.empty()) break;
    out.resize(out.size() + rowbytes);
    paint_row(cur, columns, black_is_1, reinterpret_cast<uint8_t*>(&out[out.size() - rowbytes]), rowbytes);
    ref.swap(cur);
    ++done;
  }
  if (rows > 0 && done < want) out.resize(rowbytes * (size_t)want, black_is_1 ? (char)0x00 : (char)0xFF);   // missing rows: paper
  return out;
}

}  // namespace ccitt
}  // namespace dlpdf
