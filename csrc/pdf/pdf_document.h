// PDF document: cross-reference handling, object resolution, stream filters, page tree.
#pragma once
#include <zlib.h>

#include "fast_inflate.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <unordered_set>

#include "pdf_objects.h"

namespace dlpdf {

struct XrefEntry {
  uint8_t type = 0;     // 0 free, 1 file offset, 2 in object stream
  uint64_t a = 0;       // offset | object stream number
  uint32_t b = 0;       // generation | index in object stream
};

inline long long to_int(const ObjP& o, long long dflt = 0) {
  if (!o) return dflt;
  if (o->type == Obj::Int) return o->i;
  if (o->type == Obj::Real) return (long long)llround(o->r);
  return dflt;
}

// ---------------------------------------------------------------- filters
inline std::string inflate_bytes(const uint8_t* src, size_t n, size_t hint = 0) {
  std::string out;
  out.resize(std::max<size_t>(hint, std::max<size_t>(n * 4, 4096)));
  // the in-repo decoder first (fast_inflate.h): about three times zlib's speed; whatever it refuses - a damaged or
  // truncated stream, an output that outgrows the guess - takes the tolerant zlib path below
  for (int attempt = 0; attempt < 2 && n >= 2; ++attempt) {
    const long long got = fastinflate::inflate_zlib(src, n, (uint8_t*)&out[0], out.size());
    if (got >= 0) {
      out.resize((size_t)got);
      return out;
    }
    if (attempt == 0 && out.size() < n * 64 && out.size() < (size_t)1 << 30) out.resize(std::min<size_t>(n * 64, (size_t)1 << 30));
    else break;
  }
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (inflateInit(&zs) != Z_OK) throw PdfError("inflateInit failed");
  zs.next_in = const_cast<Bytef*>(src);
  zs.avail_in = (uInt)n;
  size_t have = 0;
  int rc = Z_OK;
  while (true) {
    if (have == out.size()) out.resize(out.size() * 2);
    zs.next_out = (Bytef*)&out[have];
    zs.avail_out = (uInt)std::min<size_t>(out.size() - have, 0x40000000u);
    rc = inflate(&zs, Z_NO_FLUSH);
    have = zs.total_out;
    if (rc == Z_STREAM_END) break;
    if (rc == Z_BUF_ERROR && zs.avail_in == 0) break;  // truncated stream: keep what we have
    if (rc != Z_OK) {
      if (have > 0) break;  // tolerate trailing garbage / bad checksum
      inflateEnd(&zs);
      throw PdfError("FlateDecode: corrupt stream");
    }
  }
  inflateEnd(&zs);
  out.resize(have);
  return out;
}

inline std::string ascii_hex_decode(const uint8_t* s, size_t n) {
  std::string out;
  int hi = -1;
  for (size_t i = 0; i < n; ++i) {
    if (s[i] == '>') break;
    int v = hexval(s[i]);
    if (v < 0) continue;
    if (hi < 0) hi = v;
    else { out.push_back((char)(hi * 16 + v)); hi = -1; }
  }
  if (hi >= 0) out.push_back((char)(hi * 16));
  return out;
}

inline std::string ascii85_decode(const uint8_t* s, size_t n) {
  std::string out;
  uint32_t acc = 0;
  int cnt = 0;
  for (size_t i = 0; i < n; ++i) {
    uint8_t c = s[i];
    if (c == '~') break;
    if (is_ws(c)) continue;
    if (c == 'z' && cnt == 0) { out.append(4, '\0'); continue; }
    if (c < '!' || c > 'u') continue;
    acc = acc * 85 + (c - '!');
    if (++cnt == 5) {
      for (int k = 3; k >= 0; --k) out.push_back((char)((acc >> (8 * k)) & 0xff));
      acc = 0; cnt = 0;
    }
  }
  if (cnt > 1) {
    for (int k = cnt; k < 5; ++k) acc = acc * 85 + 84;
    for (int k = 3; k >= 5 - cnt; --k) out.push_back((char)((acc >> (8 * k)) & 0xff));
  }
  return out;
}

inline std::string runlength_decode(const uint8_t* s, size_t n) {
  std::string out;
  size_t i = 0;
  while (i < n) {
    uint8_t l = s[i++];
    if (l == 128) break;
    if (l < 128) {
      size_t c = std::min<size_t>(l + 1, n - i);
      out.append((const char*)s + i, c);
      i += c;
    } else if (i < n) {
      out.append(257 - l, (char)s[i++]);
    }
  }
  return out;
}

inline std::string lzw_decode(const uint8_t* s, size_t n, int early = 1) {
  std::string out;
  std::vector<std::string> table;
  auto reset = [&]() {
    table.clear();
    table.reserve(4096);
    for (int i = 0; i < 256; ++i) table.emplace_back(1, (char)i);
    table.emplace_back();  // 256 clear
    table.emplace_back();  // 257 eod
  };
  reset();
  int bits = 9;
  uint32_t acc = 0;
  int nb = 0;
  int prev = -1;
  for (size_t i = 0; i < n;) {
    while (nb < bits && i < n) { acc = (acc << 8) | s[i++]; nb += 8; }
    if (nb < bits) break;
    int code = (acc >> (nb - bits)) & ((1u << bits) - 1);
    nb -= bits;
    if (code == 256) { reset(); bits = 9; prev = -1; continue; }
    if (code == 257) break;
    std::string entry;
    if (code < (int)table.size()) entry = table[code];
    else if (prev >= 0) entry = table[prev] + table[prev][0];
    else break;
    out += entry;
    if (prev >= 0 && table.size() < 4096) table.push_back(table[prev] + entry[0]);
    prev = code;
    int sz = (int)table.size() + early;
    if (sz >= 2048) bits = 12;
    else if (sz >= 1024) bits = 11;
    else if (sz >= 512) bits = 10;
    else bits = 9;
  }
  return out;
}

// PNG / TIFF predictors (DecodeParms of Flate / LZW).
// PNG row filters undone straight into the destination raster (no intermediate copy): ``data`` holds ``rows`` rows of
// 1 filter byte + ``rowbytes`` bytes; ``dst`` receives rows * rowbytes bytes.  Up rows vectorise; Sub / Average / Paeth
// carry a dependency from pixel to pixel, so their loops are kept free of branches and of the "is there a left
// neighbour" test (the first bpp bytes are handled apart).
inline void png_unpredict_rows(const uint8_t* data, size_t rows, size_t rowbytes, int bpp, uint8_t* dst) {
  const size_t stride = rowbytes + 1;
  std::vector<uint8_t> zero(rowbytes, 0);
  const size_t B = (size_t)bpp;
  for (size_t y = 0; y < rows; ++y) {
    const uint8_t ft = data[y * stride];
    const uint8_t* in = data + y * stride + 1;
    uint8_t* cur = dst + y * rowbytes;
    const uint8_t* up = y ? cur - rowbytes : zero.data();
    const size_t head = B < rowbytes ? B : rowbytes;
    switch (ft) {
      case 1:
        for (size_t i = 0; i < head; ++i) cur[i] = in[i];
        if (B == 1) {
          unsigned a = cur[0];
          for (size_t i = 1; i < rowbytes; ++i) {
            a = (a + in[i]) & 255u;
            cur[i] = (uint8_t)a;
          }
        } else {
          for (size_t i = head; i < rowbytes; ++i) cur[i] = (uint8_t)(in[i] + cur[i - B]);
        }
        break;
      case 2:
        for (size_t i = 0; i < rowbytes; ++i) cur[i] = (uint8_t)(in[i] + up[i]);
        break;
      case 3:
        for (size_t i = 0; i < head; ++i) cur[i] = (uint8_t)(in[i] + (up[i] >> 1));
        if (B == 1) {
          // the left neighbour stays in a register: reading back the byte just stored costs a store-to-load forward
          // per pixel, which is most of this loop's time
          unsigned a = cur[0];
          for (size_t i = 1; i < rowbytes; ++i) {
            a = (in[i] + ((a + up[i]) >> 1)) & 255u;
            cur[i] = (uint8_t)a;
          }
        } else {
          for (size_t i = head; i < rowbytes; ++i) cur[i] = (uint8_t)(in[i] + ((cur[i - B] + up[i]) >> 1));
        }
        break;
      case 4:
        for (size_t i = 0; i < head; ++i) cur[i] = (uint8_t)(in[i] + up[i]);        // a = c = 0: the predictor is b
        if (B == 1) {
          int a = cur[0], c = up[0];
          for (size_t i = 1; i < rowbytes; ++i) {
            const int b = up[i];
            const int da = b - c, db = a - c;            // p - a, p - b; p - c is their sum
            const int pa = da < 0 ? -da : da, pb = db < 0 ? -db : db, pc = da + db < 0 ? -(da + db) : da + db;
            const int ab = pa <= pb ? a : b, pab = pa <= pb ? pa : pb;
            a = (in[i] + (pab <= pc ? ab : c)) & 255;
            cur[i] = (uint8_t)a;
            c = b;
          }
        } else {
          for (size_t i = head; i < rowbytes; ++i) {
            const int a = cur[i - B], b = up[i], c = up[i - B];
            const int da = b - c, db = a - c;
            const int pa = da < 0 ? -da : da, pb = db < 0 ? -db : db, pc = da + db < 0 ? -(da + db) : da + db;
            const int ab = pa <= pb ? a : b, pab = pa <= pb ? pa : pb;
            cur[i] = (uint8_t)(in[i] + (pab <= pc ? ab : c));
          }
        }
        break;
      default:
        memcpy(cur, in, rowbytes);
    }
  }
}

inline std::string apply_predictor(std::string data, int predictor, int colors, int bpc, int columns) {
  if (predictor <= 1) return data;
  const int bpp = std::max(1, colors * bpc / 8);
  const size_t rowbytes = ((size_t)colors * bpc * columns + 7) / 8;
  if (rowbytes == 0) return data;
  if (predictor == 2) {
    if (bpc != 8) return data;
    for (size_t off = 0; off + rowbytes <= data.size(); off += rowbytes) {
      uint8_t* r = (uint8_t*)&data[off];
      for (size_t i = bpp; i < rowbytes; ++i) r[i] = (uint8_t)(r[i] + r[i - bpp]);
    }
    return data;
  }
  // PNG predictors: each row is prefixed by a filter-type byte
  const size_t rows = data.size() / (rowbytes + 1);
  std::string out;
  out.resize(rows * rowbytes);
  png_unpredict_rows((const uint8_t*)data.data(), rows, rowbytes, bpp, (uint8_t*)&out[0]);
  return out;
}

// ---------------------------------------------------------------- document
class Document {
 public:
  Document(const uint8_t* d, size_t n) : d_(d), n_(n) {
    if (n < 8) throw PdfError("file too short to be a PDF");
    size_t hdr = find_fwd("%PDF-", 0, std::min<size_t>(n, 1024));
    if (hdr == npos) throw PdfError("missing %PDF- header");
    try {
      load_xref_chain();
    } catch (const PdfError&) {
      xref_.clear();
      trailer_ = nullptr;
    }
    if (!trailer_ || !resolve(trailer_->get("Root"))) rebuild_xref();
    if (!trailer_) throw PdfError("no trailer / catalog found");
    if (trailer_->get("Encrypt") && resolve(trailer_->get("Encrypt"))->type == Obj::Dict)
      throw PdfError("encrypted PDFs are not supported");
    build_pages();
  }

  static constexpr size_t npos = (size_t)-1;
  const uint8_t* data() const { return d_; }
  size_t size() const { return n_; }

  // Resolve indirect references (with cycle guard).
  ObjP resolve(ObjP o, int depth = 0) {
    while (o && o->type == Obj::Ref) {
      if (++depth > 32) return mk(Obj::Null);
      o = load_object(o->ref_num);
    }
    return o;
  }
  ObjP dget(const ObjP& dict, const char* key) {
    if (!dict) return nullptr;
    return resolve(dict->get(key));
  }

  ObjP load_object(int num) {
    if (num < 0 || (size_t)num >= xref_.size()) return mk(Obj::Null);
    auto it = cache_.find(num);
    if (it != cache_.end()) return it->second;
    const XrefEntry& e = xref_[num];
    ObjP res = mk(Obj::Null);
    if (e.type == 1) {
      res = parse_indirect_at((size_t)e.a, num);
    } else if (e.type == 2) {
      res = load_from_objstm((int)e.a, (int)e.b, num);
    }
    cache_[num] = res;
    return res;
  }

  // Fully decoded stream payload (all filters applied except image codecs,
  // which are reported through ``stop_filter``).
  std::string stream_data(const ObjP& st, std::string* stop_filter = nullptr) {
    if (!st || st->type != Obj::Stream) return std::string();
    const uint8_t* p;
    size_t len;
    if (st->owned) { p = (const uint8_t*)st->owned->data(); len = st->owned->size(); }
    else { p = d_ + st->stream_off; len = st->stream_len; }
    std::vector<std::string> filters;
    std::vector<ObjP> parms;
    ObjP f = dget(st, "Filter");
    ObjP dp = dget(st, "DecodeParms");
    if (!dp) dp = dget(st, "DP");
    if (f) {
      if (f->type == Obj::Name) { filters.push_back(f->s); parms.push_back(dp); }
      else if (f->type == Obj::Arr) {
        for (size_t i = 0; i < f->arr.size(); ++i) {
          ObjP fi = resolve(f->arr[i]);
          filters.push_back(fi ? fi->s : "");
          ObjP pi = (dp && dp->type == Obj::Arr && i < dp->arr.size()) ? resolve(dp->arr[i]) : nullptr;
          parms.push_back(pi);
        }
      }
    }
    std::string cur;
    bool have = false;
    for (size_t i = 0; i < filters.size(); ++i) {
      const std::string& fn = filters[i];
      const uint8_t* ip = have ? (const uint8_t*)cur.data() : p;
      size_t il = have ? cur.size() : len;
      std::string next;
      if (fn == "FlateDecode" || fn == "Fl") next = inflate_bytes(ip, il);
      else if (fn == "ASCIIHexDecode" || fn == "AHx") next = ascii_hex_decode(ip, il);
      else if (fn == "ASCII85Decode" || fn == "A85") next = ascii85_decode(ip, il);
      else if (fn == "RunLengthDecode" || fn == "RL") next = runlength_decode(ip, il);
      else if (fn == "LZWDecode" || fn == "LZW") {
        int early = 1;
        if (parms[i] && parms[i]->type == Obj::Dict) early = (int)to_int(dget(parms[i], "EarlyChange"), 1);
        next = lzw_decode(ip, il, early);
      } else {
        // image codec (DCTDecode, JPXDecode, CCITTFaxDecode, JBIG2Decode) or unknown: stop here
        if (stop_filter) *stop_filter = fn;
        if (!have) cur.assign((const char*)p, len);
        return cur;
      }
      if ((fn == "FlateDecode" || fn == "Fl" || fn == "LZWDecode" || fn == "LZW") && parms[i] &&
          parms[i]->type == Obj::Dict) {
        int pred = (int)to_int(dget(parms[i], "Predictor"), 1);
        if (pred > 1)
          next = apply_predictor(std::move(next), pred, (int)to_int(dget(parms[i], "Colors"), 1),
                                 (int)to_int(dget(parms[i], "BitsPerComponent"), 8),
                                 (int)to_int(dget(parms[i], "Columns"), 1));
      }
      cur.swap(next);
      have = true;
    }
    if (!have) cur.assign((const char*)p, len);
    return cur;
  }

  struct Page {
    ObjP dict;
    ObjP resources;
    double mediabox[4] = {0, 0, 612, 792};
    int rotate = 0;
  };
  const std::vector<Page>& pages() const { return pages_; }
  ObjP trailer() const { return trailer_; }
  ObjP catalog() { return dget(trailer_, "Root"); }
  bool was_rebuilt() const { return rebuilt_; }

 private:
  size_t find_fwd(const char* pat, size_t from, size_t to) const {
    size_t l = strlen(pat);
    if (to > n_) to = n_;
    if (to < l) return npos;
    for (size_t i = from; i + l <= to; ++i)
      if (memcmp(d_ + i, pat, l) == 0) return i;
    return npos;
  }
  size_t find_back(const char* pat, size_t from) const {
    size_t l = strlen(pat);
    if (n_ < l) return npos;
    size_t i = std::min(from, n_ - l);
    while (true) {
      if (memcmp(d_ + i, pat, l) == 0) return i;
      if (i == 0) return npos;
      --i;
    }
  }

  void ensure_xref(size_t n) {
    if (n > (size_t)8 * 1024 * 1024) throw PdfError("xref too large");
    if (xref_.size() < n) xref_.resize(n);
  }

  void load_xref_chain() {
    size_t sx = find_back("startxref", n_);
    if (sx == npos) throw PdfError("startxref not found");
    Lexer lx(d_, n_, sx + 9);
    ObjP off = lx.parse_object(false);
    if (!off || off->type != Obj::Int) throw PdfError("bad startxref");
    std::vector<size_t> visited;
    long long pos = off->i;
    while (pos >= 0 && (size_t)pos < n_) {
      if (std::find(visited.begin(), visited.end(), (size_t)pos) != visited.end()) break;
      visited.push_back((size_t)pos);
      ObjP tr = load_xref_section((size_t)pos);
      if (!tr) break;
      if (!trailer_) trailer_ = tr;
      ObjP xs = tr->get("XRefStm");
      if (xs && xs->type == Obj::Int && (size_t)xs->i < n_ &&
          std::find(visited.begin(), visited.end(), (size_t)xs->i) == visited.end()) {
        visited.push_back((size_t)xs->i);
        load_xref_section((size_t)xs->i);
      }
      ObjP prev = tr->get("Prev");
      if (!prev || !prev->is_num()) break;
      pos = (long long)prev->num();
    }
    if (!trailer_) throw PdfError("no trailer");
  }

  // Newer sections are loaded first; never overwrite an entry that is already set.
  void set_entry(size_t num, const XrefEntry& e) {
    ensure_xref(num + 1);
    if (!seen_.count(num)) { xref_[num] = e; seen_[num] = true; }
  }

  ObjP load_xref_section(size_t pos) {
    Lexer lx(d_, n_, pos);
    lx.skip_ws();
    if (lx.peek_keyword("xref")) {
      lx.read_keyword();
      while (true) {
        lx.skip_ws();
        if (lx.peek_keyword("trailer")) break;
        ObjP first = lx.parse_object(false);
        ObjP count = lx.parse_object(false);
        if (!first || !count || first->type != Obj::Int || count->type != Obj::Int) throw PdfError("bad xref subsection");
        if (count->i < 0 || count->i > 8 * 1024 * 1024) throw PdfError("bad xref count");
        for (long long k = 0; k < count->i; ++k) {
          ObjP o1 = lx.parse_object(false), o2 = lx.parse_object(false);
          std::string t = lx.read_keyword();
          if (!o1 || !o2) throw PdfError("bad xref entry");
          XrefEntry e;
          e.type = (t == "n") ? 1 : 0;
          e.a = (uint64_t)o1->i;
          e.b = (uint32_t)o2->i;
          set_entry((size_t)(first->i + k), e);
        }
      }
      lx.read_keyword();
      ObjP tr = lx.parse_object(true);
      if (!tr || tr->type != Obj::Dict) throw PdfError("bad trailer");
      return tr;
    }
    // cross-reference stream
    ObjP st = parse_indirect_at(pos, -1);
    if (!st || st->type != Obj::Stream) throw PdfError("bad xref stream");
    std::string data = stream_data(st);
    ObjP w = st->get("W");
    if (!w || w->type != Obj::Arr || w->arr.size() < 3) throw PdfError("xref stream without /W");
    int w0 = (int)to_int(w->arr[0]), w1 = (int)to_int(w->arr[1]), w2 = (int)to_int(w->arr[2]);
    if (w0 < 0 || w1 < 0 || w2 < 0 || w0 > 8 || w1 > 8 || w2 > 8) throw PdfError("bad /W");
    long long size = to_int(st->get("Size"));
    std::vector<long long> index;
    ObjP idx = st->get("Index");
    if (idx && idx->type == Obj::Arr) for (auto& v : idx->arr) index.push_back(to_int(v));
    else { index.push_back(0); index.push_back(size); }
    const size_t rec = (size_t)(w0 + w1 + w2);
    size_t p = 0;
    auto rd = [&](int wd, uint64_t dflt) -> uint64_t {
      if (wd == 0) return dflt;
      uint64_t v = 0;
      for (int k = 0; k < wd; ++k) v = (v << 8) | (uint8_t)data[p++];
      return v;
    };
    for (size_t s = 0; s + 1 < index.size(); s += 2) {
      for (long long k = 0; k < index[s + 1]; ++k) {
        if (rec == 0 || p + rec > data.size()) break;
        XrefEntry e;
        uint64_t t = rd(w0, 1);
        e.a = rd(w1, 0);
        e.b = (uint32_t)rd(w2, 0);
        e.type = (uint8_t)(t > 2 ? 0 : t);
        if (index[s] + k >= 0) set_entry((size_t)(index[s] + k), e);
      }
    }
    auto tr = mk(Obj::Dict);
    tr->dict = st->dict;
    return tr;
  }

  ObjP parse_indirect_at(size_t pos, int expect_num) {
    if (pos >= n_) return mk(Obj::Null);
    Lexer lx(d_, n_, pos);
    ObjP num = lx.parse_object(false);
    ObjP gen = lx.parse_object(false);
    std::string kw = lx.read_keyword();
    if (!num || !gen || num->type != Obj::Int || kw != "obj") throw PdfError("object header not found");
    if (expect_num >= 0 && num->i != expect_num) throw PdfError("object number mismatch");
    ObjP o = lx.parse_object(true);
    if (!o) return mk(Obj::Null);
    if (o->type == Obj::Dict && lx.peek_keyword("stream")) {
      lx.read_keyword();
      size_t p = lx.pos();
      if (p < n_ && d_[p] == '\r') ++p;
      if (p < n_ && d_[p] == '\n') ++p;
      o->type = Obj::Stream;
      o->stream_off = p;
      long long len = -1;
      ObjP l = o->get("Length");
      if (l && l->type == Obj::Ref) {
        // /Length may itself be indirect; guard against recursion into this very object
        if (l->ref_num != num->i) l = resolve(l);
      }
      if (l && l->is_num()) len = (long long)l->num();
      bool ok = len >= 0 && p + (size_t)len <= n_;
      if (ok) {
        Lexer chk(d_, n_, p + (size_t)len);
        ok = chk.peek_keyword("endstream");
      }
      if (!ok) {
        size_t e = find_fwd("endstream", p, n_);
        if (e == npos) throw PdfError("endstream not found");
        len = (long long)(e - p);
        while (len > 0 && (d_[p + len - 1] == '\n' || d_[p + len - 1] == '\r')) --len;
      }
      o->stream_len = (size_t)len;
    }
    return o;
  }

  ObjP load_from_objstm(int stm_num, int index, int want_num) {
    ObjP stm = load_object(stm_num);
    if (!stm || stm->type != Obj::Stream) return mk(Obj::Null);
    auto it = objstm_cache_.find(stm_num);
    if (it == objstm_cache_.end()) {
      auto data = std::make_shared<std::string>(stream_data(stm));
      it = objstm_cache_.emplace(stm_num, data).first;
    }
    const std::string& data = *it->second;
    long long n = to_int(dget(stm, "N")), first = to_int(dget(stm, "First"));
    if (n <= 0 || first < 0 || (size_t)first > data.size()) return mk(Obj::Null);
    Lexer hdr((const uint8_t*)data.data(), (size_t)first);
    for (long long k = 0; k < n; ++k) {
      ObjP on = hdr.parse_object(false), off = hdr.parse_object(false);
      if (!on || !off) break;
      if (on->i == want_num) {
        size_t pos = (size_t)first + (size_t)off->i;
        if (pos >= data.size()) return mk(Obj::Null);
        Lexer body((const uint8_t*)data.data(), data.size(), pos);
        ObjP o = body.parse_object(true);
        return o ? o : mk(Obj::Null);
      }
    }
    (void)index;
    return mk(Obj::Null);
  }

  // Damaged / missing xref: scan the file for "n g obj" headers and trailers.
  void rebuild_xref() {
    rebuilt_ = true;
    xref_.clear();
    seen_.clear();
    cache_.clear();
    trailer_ = nullptr;
    for (size_t i = 0; i + 3 < n_; ++i) {
      if (d_[i] == 'o' && d_[i + 1] == 'b' && d_[i + 2] == 'j' && (i + 3 >= n_ || is_ws(d_[i + 3]) || is_delim(d_[i + 3]))) {
        // walk back over "<num> <gen> "
        size_t j = i;
        auto back_ws = [&]() { while (j > 0 && is_ws(d_[j - 1])) --j; };
        auto back_digits = [&]() { size_t s = j; while (j > 0 && d_[j - 1] >= '0' && d_[j - 1] <= '9') --j; return s - j; };
        back_ws();
        if (!back_digits()) continue;
        back_ws();
        size_t e = j;
        if (!back_digits()) continue;
        long long num = std::stoll(std::string((const char*)d_ + j, e - j));
        if (num < 0 || num > 8 * 1024 * 1024) continue;
        ensure_xref((size_t)num + 1);
        XrefEntry en; en.type = 1; en.a = j; en.b = 0;
        xref_[(size_t)num] = en;  // later definitions win (incremental updates)
      }
    }
    // trailers: last one with /Root wins; also accept xref-stream dictionaries
    size_t pos = 0;
    while ((pos = find_fwd("trailer", pos, n_)) != npos) {
      try {
        Lexer lx(d_, n_, pos + 7);
        ObjP tr = lx.parse_object(true);
        if (tr && tr->type == Obj::Dict && tr->get("Root")) trailer_ = tr;
      } catch (const PdfError&) {}
      pos += 7;
    }
    if (!trailer_) {
      for (size_t num = 0; num < xref_.size(); ++num) {
        if (xref_[num].type != 1) continue;
        try {
          ObjP o = load_object((int)num);
          if (o && (o->type == Obj::Dict || o->type == Obj::Stream)) {
            ObjP ty = o->get("Type");
            if (ty && ty->type == Obj::Name && ty->s == "Catalog") {
              trailer_ = mk(Obj::Dict);
              auto r = mk(Obj::Ref); r->ref_num = (int)num;
              trailer_->dict["Root"] = r;
            } else if (ty && ty->type == Obj::Name && ty->s == "XRef" && o->get("Root")) {
              trailer_ = mk(Obj::Dict);
              trailer_->dict = o->dict;
              break;
            }
          }
        } catch (const PdfError&) {}
      }
    }
  }

  void build_pages() {
    ObjP cat = catalog();
    if (!cat || cat->type != Obj::Dict) throw PdfError("catalog missing");
    ObjP root = dget(cat, "Pages");
    if (!root) throw PdfError("page tree missing");
    Page inh;
    std::unordered_set<const Obj*> on_path;
    size_t visited = 0;
    walk_pages(root, inh, 0, on_path, visited);
  }

  void read_box(const ObjP& a, double* box) {
    if (!a || a->type != Obj::Arr || a->arr.size() < 4) return;
    double v[4];
    for (int k = 0; k < 4; ++k) { ObjP e = resolve(a->arr[k]); v[k] = (e && e->is_num()) ? e->num() : 0; }
    box[0] = std::min(v[0], v[2]); box[1] = std::min(v[1], v[3]);
    box[2] = std::max(v[0], v[2]); box[3] = std::max(v[1], v[3]);
  }

  // Depth-first over /Kids.  A node already on the current path is a cycle and is skipped; the total node budget
  // bounds trees that repeat a subtree many times (A.Kids=[B,B], B.Kids=[C,C], ... doubles per level).
  void walk_pages(const ObjP& node, Page inh, int depth, std::unordered_set<const Obj*>& on_path, size_t& visited) {
    if (!node || node->type != Obj::Dict || depth > 64) return;
    if (pages_.size() > 200000) throw PdfError("too many pages");
    if (++visited > 1000000) throw PdfError("page tree too large");
    if (!on_path.insert(node.get()).second) return;
    struct Pop { std::unordered_set<const Obj*>& s; const Obj* p; ~Pop() { s.erase(p); } } pop{on_path, node.get()};
    ObjP res = dget(node, "Resources");
    if (res && res->type == Obj::Dict) inh.resources = res;
    ObjP mb = dget(node, "MediaBox");
    if (mb) read_box(mb, inh.mediabox);
    ObjP rot = dget(node, "Rotate");
    if (rot && rot->is_num()) inh.rotate = (((int)rot->num() % 360) + 360) % 360;
    ObjP kids = dget(node, "Kids");
    ObjP type = dget(node, "Type");
    bool is_page = (type && type->type == Obj::Name && type->s == "Page") || (!kids);
    if (!is_page && kids && kids->type == Obj::Arr) {
      for (auto& k : kids->arr) {
        ObjP kid = resolve(k);
        walk_pages(kid, inh, depth + 1, on_path, visited);
      }
      return;
    }
    ObjP cb = dget(node, "CropBox");
    if (cb) {
      double b[4] = {0, 0, 0, 0};
      read_box(cb, b);
      if (b[2] > b[0] && b[3] > b[1]) memcpy(inh.mediabox, b, sizeof(b));
    }
    inh.dict = node;
    pages_.push_back(inh);
  }

  const uint8_t* d_;
  size_t n_;
  std::vector<XrefEntry> xref_;
  std::unordered_map<size_t, bool> seen_;
  std::unordered_map<int, ObjP> cache_;
  std::unordered_map<int, std::shared_ptr<std::string>> objstm_cache_;
  ObjP trailer_;
  std::vector<Page> pages_;
  bool rebuilt_ = false;
};

}  // namespace dlpdf
