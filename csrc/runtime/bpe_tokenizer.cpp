// Byte-level BPE vocabulary (GPT-2 / RoBERTa family, the vocabulary TrOCR-sized decoders use) in C++.
//
// The reference reaches the Rust `tokenizers` wheel through its recognizers
// (requirements.txt:293); this is the in-repo native replacement.  Two vocabularies:
//   * the built-in raw byte vocabulary: ids 0..3 = <s>, <pad>, </s>, <unk>; ids 4..259 = bytes 0..255
//     (what models/trocr.py uses when no vocab file is given);
//   * a loaded `vocab.json` (token -> id, tokens written in the printable byte alphabet) with an optional
//     `merges.txt` (ranked pair merges) for encoding.
// Decoding ids -> UTF-8 is the serving hot path (one call per recognised batch, rows packed into one
// buffer); encoding text -> ids is needed only to build training targets.
//
// C ABI (ctypes, runtime/tokenizer.py):
//   dlbpe_create_bytes() / dlbpe_create(vocab_json, n, merges_txt, m) -> handle (nullptr on parse error)
//   dlbpe_destroy, dlbpe_vocab_size, dlbpe_special(handle, which)
//   dlbpe_decode_batch(handle, ids[rows*cols], rows, cols, stop_at_eos, out, cap, offsets[rows+1]) -> bytes or -needed
//   dlbpe_encode(handle, utf8, n, out_ids, cap) -> count or -needed
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

// The printable alphabet: bytes '!'..'~', 0xA1..0xAC, 0xAE..0xFF stand for themselves (as code points);
// the remaining 68 bytes are mapped to U+0100 + n in increasing byte order.
struct ByteAlphabet {
  uint32_t to_cp[256];
  std::unordered_map<uint32_t, uint8_t> to_byte;
  ByteAlphabet() {
    int n = 0;
    for (int b = 0; b < 256; ++b) {
      const bool keep = (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
      to_cp[b] = keep ? (uint32_t)b : 0x100u + (uint32_t)(n++);
      to_byte[to_cp[b]] = (uint8_t)b;
    }
  }
};

const ByteAlphabet& alphabet() {
  static ByteAlphabet a;
  return a;
}

void append_utf8(std::string& s, uint32_t cp) {
  if (cp < 0x80) s.push_back((char)cp);
  else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

// Decodes one code point at s[i]; advances i. Malformed input yields U+FFFD and advances one byte.
uint32_t next_cp(const std::string& s, size_t& i) {
  const unsigned char c = (unsigned char)s[i];
  int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
  if (len == 0 || i + len > s.size()) { ++i; return 0xFFFD; }
  uint32_t cp = len == 1 ? c : c & (0xFF >> (len + 1));
  for (int k = 1; k < len; ++k) {
    const unsigned char d = (unsigned char)s[i + k];
    if ((d & 0xC0) != 0x80) { ++i; return 0xFFFD; }
    cp = (cp << 6) | (d & 0x3F);
  }
  i += len;
  return cp;
}

// Token text in the printable alphabet -> raw bytes. Code points outside the alphabet (special tokens such
// as "<s>" are all inside it; anything else is kept as its UTF-8) pass through unchanged.
std::string alphabet_to_bytes(const std::string& tok) {
  std::string out;
  const auto& a = alphabet();
  for (size_t i = 0; i < tok.size();) {
    const uint32_t cp = next_cp(tok, i);
    auto it = a.to_byte.find(cp);
    if (it != a.to_byte.end()) out.push_back((char)it->second);
    else append_utf8(out, cp);
  }
  return out;
}

std::string bytes_to_alphabet(const char* p, size_t n) {
  std::string out;
  const auto& a = alphabet();
  for (size_t i = 0; i < n; ++i) append_utf8(out, a.to_cp[(unsigned char)p[i]]);
  return out;
}

// ---- a JSON reader for exactly one shape: {"token": int, ...} (strings with the standard escapes)
struct JsonCursor {
  const char* p;
  const char* e;
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p; }
  bool lit(char c) { ws(); if (p < e && *p == c) { ++p; return true; } return false; }
  static int hex(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
  bool u16(uint32_t& v) {
    if (e - p < 4) return false;
    v = 0;
    for (int k = 0; k < 4; ++k) { const int h = hex(p[k]); if (h < 0) return false; v = v * 16 + h; }
    p += 4;
    return true;
  }
  bool str(std::string& out) {
    ws();
    if (p >= e || *p != '"') return false;
    ++p;
    out.clear();
    while (p < e && *p != '"') {
      if (*p != '\\') { out.push_back(*p++); continue; }
      if (++p >= e) return false;
      const char c = *p++;
      switch (c) {
        case 'n': out.push_back('\n'); break;
        case 't': out.push_back('\t'); break;
        case 'r': out.push_back('\r'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'u': {
          uint32_t v;
          if (!u16(v)) return false;
          if (v >= 0xD800 && v < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {   // surrogate pair
            p += 2;
            uint32_t lo;
            if (!u16(lo)) return false;
            v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
          }
          append_utf8(out, v);
          break;
        }
        default: out.push_back(c);   // \" \\ \/
      }
    }
    if (p >= e) return false;
    ++p;
    return true;
  }
  bool integer(long long& v) {
    ws();
    bool neg = false;
    if (p < e && *p == '-') { neg = true; ++p; }
    if (p >= e || *p < '0' || *p > '9') return false;
    v = 0;
    while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (*p++ - '0'); if (v > (1LL << 40)) return false; }
    if (neg) v = -v;
    return true;
  }
};

struct PairHash {
  size_t operator()(const std::pair<int, int>& p) const { return ((size_t)(uint32_t)p.first << 32) ^ (uint32_t)p.second; }
};

struct Bpe {
  std::vector<std::string> id_to_bytes;            // raw bytes each id expands to ("" for specials / holes)
  std::vector<uint8_t> is_special;
  std::unordered_map<std::string, int> tok_to_id;  // keyed by raw bytes
  std::unordered_map<std::pair<int, int>, std::pair<int, int>, PairHash> merges;   // (a,b) -> (rank, merged id)
  int bos = 0, pad = 1, eos = 2, unk = 3;
  bool raw_bytes = false;
};

Bpe* make_raw() {
  auto* b = new Bpe();
  b->raw_bytes = true;
  b->id_to_bytes.resize(260);
  b->is_special.assign(260, 0);
  for (int i = 0; i < 4; ++i) b->is_special[i] = 1;
  for (int i = 0; i < 256; ++i) {
    b->id_to_bytes[4 + i] = std::string(1, (char)i);
    b->tok_to_id[b->id_to_bytes[4 + i]] = 4 + i;
  }
  return b;
}

int find_special(const std::unordered_map<std::string, int>& printable, const char* name, int dflt) {
  auto it = printable.find(name);
  return it == printable.end() ? dflt : it->second;
}

Bpe* make_from_files(const char* vj, size_t vn, const char* mt, size_t mn) {
  JsonCursor c{vj, vj + vn};
  if (!c.lit('{')) return nullptr;
  std::unordered_map<std::string, int> printable;
  long long max_id = -1;
  if (!c.lit('}')) {
    for (;;) {
      std::string k;
      long long v;
      if (!c.str(k) || !c.lit(':') || !c.integer(v) || v < 0 || v > 4'000'000) return nullptr;
      printable[k] = (int)v;
      max_id = std::max(max_id, v);
      if (c.lit(',')) continue;
      if (c.lit('}')) break;
      return nullptr;
    }
  }
  if (max_id < 0) return nullptr;
  auto* b = new Bpe();
  b->id_to_bytes.resize((size_t)max_id + 1);
  b->is_special.assign((size_t)max_id + 1, 0);
  for (const auto& kv : printable) {
    const std::string& t = kv.first;
    const bool special = t.size() >= 3 && t.front() == '<' && t.back() == '>' &&
                         (t == "<s>" || t == "</s>" || t == "<pad>" || t == "<unk>" || t == "<mask>" || t == "<|endoftext|>");
    if (special) { b->is_special[kv.second] = 1; continue; }
    b->id_to_bytes[kv.second] = alphabet_to_bytes(t);
    b->tok_to_id[b->id_to_bytes[kv.second]] = kv.second;
  }
  b->bos = find_special(printable, "<s>", find_special(printable, "<|endoftext|>", 0));
  b->pad = find_special(printable, "<pad>", b->bos);
  b->eos = find_special(printable, "</s>", find_special(printable, "<|endoftext|>", 2));
  b->unk = find_special(printable, "<unk>", b->eos);
  // merges.txt: "#version" header optional, then "left right" per line in rank order
  int rank = 0;
  const char* p = mt;
  const char* e = mt + mn;
  while (p && p < e) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
    const char* le = nl ? nl : e;
    std::string line(p, le);
    p = nl ? nl + 1 : e;
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    const size_t sp = line.find(' ');
    if (sp == std::string::npos) continue;
    const std::string l = alphabet_to_bytes(line.substr(0, sp)), r = alphabet_to_bytes(line.substr(sp + 1));
    auto il = b->tok_to_id.find(l), ir = b->tok_to_id.find(r), im = b->tok_to_id.find(l + r);
    if (il != b->tok_to_id.end() && ir != b->tok_to_id.end() && im != b->tok_to_id.end())
      b->merges.emplace(std::make_pair(il->second, ir->second), std::make_pair(rank, im->second));
    ++rank;
  }
  return b;
}

// Pre-tokenisation: a word is an optional leading space followed by a run of one class (letters, digits,
// other non-space), or a run of whitespace. This is the split the GPT-2 regex produces for ASCII text and
// a close approximation elsewhere (non-ASCII bytes count as letters); contractions are not special-cased.
enum Cls { SPACE, LETTER, DIGIT, OTHER };
Cls cls_of(unsigned char c) {
  if (c == ' ' || c == '\n' || c == '\t' || c == '\r') return SPACE;
  if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c >= 0x80) return LETTER;
  if (c >= '0' && c <= '9') return DIGIT;
  return OTHER;
}

void split_words(const char* s, size_t n, std::vector<std::pair<size_t, size_t>>& words) {
  size_t i = 0;
  while (i < n) {
    size_t st = i;
    if (s[i] == ' ' && i + 1 < n && cls_of((unsigned char)s[i + 1]) != SPACE) ++i;   // the space glues to the word
    const Cls c = cls_of((unsigned char)s[i]);
    if (c == SPACE) {
      while (i < n && cls_of((unsigned char)s[i]) == SPACE) ++i;
      if (i < n && i - st > 1 && s[i - 1] == ' ') --i;     // leave the last space for the next word
    } else {
      while (i < n && cls_of((unsigned char)s[i]) == c) ++i;
    }
    words.emplace_back(st, i);
  }
}

void encode_word(const Bpe& b, const char* s, size_t n, std::vector<int>& out) {
  std::vector<int> sym;
  sym.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    auto it = b.tok_to_id.find(std::string(1, s[i]));
    sym.push_back(it == b.tok_to_id.end() ? b.unk : it->second);
  }
  while (sym.size() > 1 && !b.merges.empty()) {
    int best_rank = std::numeric_limits<int>::max(), best_id = -1;
    size_t best_i = 0;
    for (size_t i = 0; i + 1 < sym.size(); ++i) {
      auto it = b.merges.find({sym[i], sym[i + 1]});
      if (it != b.merges.end() && it->second.first < best_rank) { best_rank = it->second.first; best_id = it->second.second; best_i = i; }
    }
    if (best_id < 0) break;
    // merge every occurrence of the winning pair, left to right (the rule the reference tokenizers apply)
    const int a0 = sym[best_i], a1 = sym[best_i + 1];
    std::vector<int> nxt;
    nxt.reserve(sym.size());
    for (size_t i = 0; i < sym.size();) {
      if (i + 1 < sym.size() && sym[i] == a0 && sym[i + 1] == a1) { nxt.push_back(best_id); i += 2; }
      else nxt.push_back(sym[i++]);
    }
    sym.swap(nxt);
  }
  out.insert(out.end(), sym.begin(), sym.end());
}

}  // namespace

extern "C" {

void* dlbpe_create_bytes() { return make_raw(); }

void* dlbpe_create(const char* vocab_json, long long n, const char* merges_txt, long long m) {
  if (!vocab_json || n <= 0) return nullptr;
  return make_from_files(vocab_json, (size_t)n, merges_txt, merges_txt ? (size_t)std::max<long long>(m, 0) : 0);
}

void dlbpe_destroy(void* h) { delete static_cast<Bpe*>(h); }

int dlbpe_vocab_size(void* h) { return h ? (int)static_cast<Bpe*>(h)->id_to_bytes.size() : 0; }

// which: 0 bos, 1 pad, 2 eos, 3 unk
int dlbpe_special(void* h, int which) {
  if (!h) return -1;
  const Bpe& b = *static_cast<Bpe*>(h);
  return which == 0 ? b.bos : which == 1 ? b.pad : which == 2 ? b.eos : which == 3 ? b.unk : -1;
}

long long dlbpe_decode_batch(void* h, const int32_t* ids, int rows, int cols, int stop_at_eos, char* out,
                             long long cap, long long* offsets) {
  if (!h || !ids || rows < 0 || cols < 0 || !offsets) return -1;
  const Bpe& b = *static_cast<Bpe*>(h);
  const long long V = (long long)b.id_to_bytes.size();
  long long pos = 0;
  for (int r = 0; r < rows; ++r) {
    offsets[r] = pos;
    const int32_t* row = ids + (long long)r * cols;
    for (int c = 0; c < cols; ++c) {
      const int t = row[c];
      if (stop_at_eos && t == b.eos) break;
      if (t < 0 || t >= V || b.is_special[t]) continue;
      const std::string& s = b.id_to_bytes[t];
      if (out && pos + (long long)s.size() <= cap) memcpy(out + pos, s.data(), s.size());
      pos += (long long)s.size();
    }
  }
  offsets[rows] = pos;
  return (out && pos <= cap) ? pos : -pos - 1;    // negative: -(needed)-1, nothing usable was written past cap
}

long long dlbpe_encode(void* h, const char* utf8, long long n, int32_t* out_ids, long long cap) {
  if (!h || (!utf8 && n > 0) || n < 0) return -1;
  const Bpe& b = *static_cast<Bpe*>(h);
  std::vector<int> ids;
  if (b.raw_bytes) {
    ids.reserve((size_t)n);
    for (long long i = 0; i < n; ++i) ids.push_back(4 + (unsigned char)utf8[i]);
  } else {
    std::vector<std::pair<size_t, size_t>> words;
    split_words(utf8, (size_t)n, words);
    for (const auto& w : words) encode_word(b, utf8 + w.first, w.second - w.first, ids);
  }
  if (out_ids && (long long)ids.size() <= cap) {
    for (size_t i = 0; i < ids.size(); ++i) out_ids[i] = ids[i];
    return (long long)ids.size();
  }
  return -(long long)ids.size() - 1;
}

// Printable-alphabet form of raw bytes (for writing vocab.json / merges.txt from Python tools and tests).
long long dlbpe_printable(const char* bytes, long long n, char* out, long long cap) {
  const std::string s = bytes_to_alphabet(bytes, (size_t)std::max<long long>(n, 0));
  if (out && (long long)s.size() <= cap) { memcpy(out, s.data(), s.size()); return (long long)s.size(); }
  return -(long long)s.size() - 1;
}

}  // extern "C"
