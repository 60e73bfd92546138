// Bit reader, Huffman symbol decode and the subsequence decoder of the self-synchronising JPEG path, shared by the
// GPU kernels (jpeg_decode.hip) and a serial host emulation of the same pipeline (csrc/pdf/jpeg_sync_host.h) that the
// CPU tests check against libjpeg.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define JPEG_HD __host__ __device__ __forceinline__
#else
#define JPEG_HD inline
#endif

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
  // restart-less streams only (self-synchronising decode)
  int32_t* dcdiff;            // [n luma blocks] DC differences in decode order, zero-filled by the caller
  uint32_t* sub_state;        // [n_sub] state at the END of each subsequence: d | k << 16 | b << 23
  int32_t* sub_count;         // [n_sub] blocks completed inside each subsequence
  const int32_t* sub_off;     // [n_sub] exclusive prefix sum of sub_count (filled between kernels b and d)
  int32_t n_sub, scan_off, total_blocks, pad2;
};
static_assert(sizeof(JpegImageDesc) == 136, "ABI");

#define JPEG_SUB_BYTES 128      // raw bytes per subsequence
#define JPEG_SUB_CHUNK 256      // subsequences per workgroup of the synchronisation kernel

JPEG_HD int jpeg_imin(int a, int b) { return a < b ? a : b; }
JPEG_HD int jpeg_imax(int a, int b) { return a > b ? a : b; }

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t acc;   // MSB-aligned
  int nbits;
  bool hit_marker;

  JPEG_HD void init(const uint8_t* s, const uint8_t* e) {
    p = s; end = e; acc = 0; nbits = 0; hit_marker = false;
  }
  // guarantee >= 32 valid bits (zero-extended past the end of the segment)
  JPEG_HD void refill() {
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
  JPEG_HD uint32_t peek(int n) const { return (uint32_t)(acc >> (64 - n)); }
  JPEG_HD void skip(int n) { acc <<= n; nbits -= n; }
};

JPEG_HD int huff_decode(BitReader& br, const uint16_t* lut, const JpegTables* __restrict__ tb, int t) {
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

JPEG_HD int receive_extend(BitReader& br, int s) {
  if (s == 0) return 0;
  int v = (int)br.peek(s);
  br.skip(s);
  return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

// ---------------------------------------------------------------------------------------------------------
// Self-synchronising decode of restart-less scans.
//
// A position inside the scan is (subsequence j, d): d = DATA bits (stuffing bytes removed) after the first raw byte
// of subsequence j.  A decoder entering subsequence j starts its bit reader at that byte (skipping a stuffing zero
// whose 0xFF is the last byte of subsequence j-1) and skips d bits; the overshoot of the previous subsequence's
// last symbol is at most one code word plus its magnitude bits (27), so d always fits one refill.
struct SubState {
  int d;   // data-bit offset of the next code word inside the subsequence
  int k;   // next coefficient index of the block in progress (0: a DC code comes next)
  int b;   // index of the block in progress inside its MCU
};
JPEG_HD uint32_t sub_pack(const SubState& s) { return (uint32_t)s.d | ((uint32_t)s.k << 16) | ((uint32_t)s.b << 23); }
JPEG_HD SubState sub_unpack(uint32_t v) { return SubState{(int)(v & 0xffffu), (int)((v >> 16) & 127u), (int)(v >> 23)}; }

JPEG_HD int sub_raw_start(const JpegImageDesc& d, int j) { return d.scan_off + j * JPEG_SUB_BYTES; }

// Stuffing zeros (a 0x00 that follows a 0xFF) among the raw bytes of subsequence j.
JPEG_HD int sub_stuffing(const JpegImageDesc& d, int j) {
  const int s = sub_raw_start(d, j), e = jpeg_imin(s + JPEG_SUB_BYTES, d.scan_end);
  int n = 0;
  uint32_t prev = s > d.scan_off ? d.data[s - 1] : 0u;
  for (int q = s; q < e; ++q) {
    const uint32_t c = d.data[q];
    n += (c == 0u && prev == 0xFFu) ? 1 : 0;
    prev = c;
  }
  return n;
}

// Decode subsequence j from state st (updated to the state at its end).  Returns the number of blocks completed.
// WRITE: also store coefficients; blk0 = global index (decode order, all components) of the block in progress at entry.
template <bool WRITE>
JPEG_HD int sub_decode(const JpegImageDesc& d, const int j, SubState& st, const int stuffing, int blk0,
                                          const uint16_t (*s_lut)[JPEG_LUT_SIZE], const uint16_t* s_qt, const uint8_t* s_zz) {
  const JpegTables* tb = d.tables;
  int s = sub_raw_start(d, j);
  const int e = jpeg_imin(s + JPEG_SUB_BYTES, d.scan_end);
  const int nbits_sub = 8 * (e - s - stuffing);           // data bits that belong to this subsequence
  if (s > d.scan_off && d.data[s - 1] == 0xFF && d.data[s] == 0x00) ++s;
  BitReader br;
  br.init(d.data + s, d.data + d.scan_end);
  int consumed = st.d;
  br.refill();
  br.skip(st.d);
  const int yblocks = d.hs0 * d.vs0;
  const int nblk = d.ncomp == 1 ? 1 : yblocks + 2;
  int k = st.k, b = st.b, done = 0;
  // position of the block in progress (only when writing)
  int mcu = 0, mx = 0, my = 0;
  if (WRITE) {
    mcu = blk0 / nblk;
    my = mcu / d.mcus_x;
    mx = mcu - my * d.mcus_x;
  }
  while (consumed < nbits_sub) {
    const bool luma = b < yblocks;
    const bool dc = k == 0;
    br.refill();
    // One code path for DC and AC symbols: lanes of a wave sit at different coefficient indices, and a branch per
    // symbol kind would run both kinds for every lane.  A DC symbol is (run 0, size), so the AC arithmetic covers it.
    const int t = (luma ? 0 : 2) + (dc ? 0 : 1);
    const uint32_t e0 = s_lut[t][br.peek(JPEG_LUT_BITS)];
    int rs, len;
    if (e0) { len = (int)(e0 >> 8); rs = (int)(e0 & 0xff); }
    else {
      len = JPEG_LUT_BITS + 1;
      int code = (int)br.peek(len);
      while (len <= 16 && code > tb->maxcode[t][len]) { ++len; code = (int)br.peek(len); }
      if (len > 16) { len = 16; rs = 0; }
      else rs = tb->vals[t][(code + tb->valoff[t][len]) & 0xff];
    }
    br.skip(len);
    const int r = rs >> 4, sz = rs & 15;
    const int v = receive_extend(br, sz);              // >= 16 bits are left after the code word: no refill needed
    consumed += len + sz;
    if (dc) {
      if (WRITE && luma && blk0 + done < d.total_blocks) d.dcdiff[(long long)mcu * yblocks + b] = v;
      k = 1;
    } else if (sz == 0) {
      k = (r == 15) ? k + 16 : 64;                      // ZRL or end of block
    } else {
      k += r;
      if (WRITE && luma && k < 64 && blk0 + done < d.total_blocks) {
        const int bx = mx * d.hs0 + (b % d.hs0), by = my * d.vs0 + (b / d.hs0);
        const int z = s_zz[k];
        const int q = v * (int)s_qt[z];
        d.coef[((long long)by * d.Wb + bx) * 64 + z] = (int16_t)jpeg_imax(-32768, jpeg_imin(32767, q));
      }
      ++k;
    }
    if (k >= 64) {                                       // block complete
      k = 0;
      ++done;
      if (++b == nblk) {
        b = 0;
        if (WRITE) { ++mcu; if (++mx == d.mcus_x) { mx = 0; ++my; } }
      }
    }
  }
  st.d = consumed - nbits_sub;
  st.k = k;
  st.b = b;
  return done;
}

