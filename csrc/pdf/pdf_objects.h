// Minimal PDF object model + lexer/parser for the native reader.
// Written from the PDF 1.7 specification (ISO 32000-1); replaces PyMuPDF in the
// reference's stack (services/ocr_service/requirements.txt:214).
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace dlpdf {

struct PdfError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct Obj;
using ObjP = std::shared_ptr<Obj>;

struct Obj {
  enum Type { Null, Bool, Int, Real, Str, Name, Arr, Dict, Ref, Stream } type = Null;
  bool b = false;
  long long i = 0;
  double r = 0.0;
  std::string s;                        // Str / Name payload (raw bytes)
  std::vector<ObjP> arr;
  std::map<std::string, ObjP> dict;     // Dict and Stream dictionary
  int ref_num = 0, ref_gen = 0;
  size_t stream_off = 0, stream_len = 0;  // raw (still filtered) stream bytes inside the file buffer
  std::shared_ptr<std::string> owned;     // stream payload when it does not live in the file buffer

  bool is_num() const { return type == Int || type == Real; }
  double num() const { return type == Int ? (double)i : r; }
  ObjP get(const std::string& k) const {
    auto it = dict.find(k);
    return it == dict.end() ? nullptr : it->second;
  }
};

inline ObjP mk(Obj::Type t) {
  auto o = std::make_shared<Obj>();
  o->type = t;
  return o;
}

inline bool is_ws(uint8_t c) { return c == 0 || c == 9 || c == 10 || c == 12 || c == 13 || c == 32; }
inline bool is_delim(uint8_t c) {
  return c == '(' || c == ')' || c == '<' || c == '>' || c == '[' || c == ']' || c == '{' || c == '}' || c == '/' ||
         c == '%';
}
inline int hexval(uint8_t c) {
  if (c >= '0' && c <= '9') return c - '0';
  if (c >= 'a' && c <= 'f') return c - 'a' + 10;
  if (c >= 'A' && c <= 'F') return c - 'A' + 10;
  return -1;
}

// Recursive-descent parser over a byte buffer.  Used for file bodies, object
// streams and (with keyword tokens) content streams.
class Lexer {
 public:
  Lexer(const uint8_t* d, size_t n, size_t pos = 0) : d_(d), n_(n), pos_(pos) {}
  size_t pos() const { return pos_; }
  void seek(size_t p) { pos_ = p; }
  bool eof() const { return pos_ >= n_; }
  const uint8_t* data() const { return d_; }
  size_t size() const { return n_; }

  void skip_ws() {
    while (pos_ < n_) {
      uint8_t c = d_[pos_];
      if (is_ws(c)) {
        ++pos_;
      } else if (c == '%') {
        while (pos_ < n_ && d_[pos_] != '\n' && d_[pos_] != '\r') ++pos_;
      } else {
        break;
      }
    }
  }

  // Reads a bare keyword / operator token (obj, endobj, stream, R, BT, Tj ...).
  std::string read_keyword() {
    skip_ws();
    size_t s = pos_;
    while (pos_ < n_ && !is_ws(d_[pos_]) && !is_delim(d_[pos_])) ++pos_;
    return std::string((const char*)d_ + s, pos_ - s);
  }

  bool peek_keyword(const char* kw) {
    skip_ws();
    size_t l = strlen(kw);
    if (pos_ + l > n_) return false;
    if (memcmp(d_ + pos_, kw, l) != 0) return false;
    if (pos_ + l < n_ && !is_ws(d_[pos_ + l]) && !is_delim(d_[pos_ + l])) return false;
    return true;
  }

  // Parses one object.  Returns nullptr when the next token is a keyword
  // (operator); the keyword is left unread.  ``allow_ref`` enables "n g R".
  ObjP parse_object(bool allow_ref = true, int depth = 0) {
    if (depth > 64) throw PdfError("object nesting too deep");
    skip_ws();
    if (pos_ >= n_) return nullptr;
    uint8_t c = d_[pos_];
    if (c == '/') return parse_name();
    if (c == '(') return parse_literal_string();
    if (c == '[') {
      ++pos_;
      auto a = mk(Obj::Arr);
      while (true) {
        skip_ws();
        if (pos_ >= n_) throw PdfError("unterminated array");
        if (d_[pos_] == ']') { ++pos_; break; }
        auto e = parse_object(allow_ref, depth + 1);
        if (!e) {  // stray keyword inside array: skip it
          std::string kw = read_keyword();
          if (kw.empty()) ++pos_;
          continue;
        }
        a->arr.push_back(e);
      }
      return a;
    }
    if (c == '<') {
      if (pos_ + 1 < n_ && d_[pos_ + 1] == '<') return parse_dict(allow_ref, depth);
      return parse_hex_string();
    }
    if (c == '+' || c == '-' || c == '.' || (c >= '0' && c <= '9')) return parse_number(allow_ref);
    if (c == ')' || c == '>' || c == ']' || c == '}' || c == '{') {
      return nullptr;
    }
    // keyword: true / false / null are objects, everything else is an operator
    size_t save = pos_;
    std::string kw = read_keyword();
    if (kw == "true") { auto o = mk(Obj::Bool); o->b = true; return o; }
    if (kw == "false") { auto o = mk(Obj::Bool); o->b = false; return o; }
    if (kw == "null") return mk(Obj::Null);
    pos_ = save;
    return nullptr;
  }

 private:
  ObjP parse_name() {
    ++pos_;
    auto o = mk(Obj::Name);
    while (pos_ < n_ && !is_ws(d_[pos_]) && !is_delim(d_[pos_])) {
      uint8_t c = d_[pos_];
      if (c == '#' && pos_ + 2 < n_ && hexval(d_[pos_ + 1]) >= 0 && hexval(d_[pos_ + 2]) >= 0) {
        o->s.push_back((char)(hexval(d_[pos_ + 1]) * 16 + hexval(d_[pos_ + 2])));
        pos_ += 3;
      } else {
        o->s.push_back((char)c);
        ++pos_;
      }
    }
    return o;
  }

  ObjP parse_literal_string() {
    ++pos_;
    auto o = mk(Obj::Str);
    int depth = 1;
    while (pos_ < n_) {
      uint8_t c = d_[pos_++];
      if (c == '\\') {
        if (pos_ >= n_) break;
        uint8_t e = d_[pos_++];
        switch (e) {
          case 'n': o->s.push_back('\n'); break;
          case 'r': o->s.push_back('\r'); break;
          case 't': o->s.push_back('\t'); break;
          case 'b': o->s.push_back('\b'); break;
          case 'f': o->s.push_back('\f'); break;
          case '\r': if (pos_ < n_ && d_[pos_] == '\n') ++pos_; break;
          case '\n': break;
          default:
            if (e >= '0' && e <= '7') {
              int v = e - '0';
              for (int k = 0; k < 2 && pos_ < n_ && d_[pos_] >= '0' && d_[pos_] <= '7'; ++k) v = v * 8 + (d_[pos_++] - '0');
              o->s.push_back((char)(v & 0xff));
            } else {
              o->s.push_back((char)e);
            }
        }
      } else if (c == '(') {
        ++depth;
        o->s.push_back('(');
      } else if (c == ')') {
        if (--depth == 0) return o;
        o->s.push_back(')');
      } else {
        o->s.push_back((char)c);
      }
    }
    throw PdfError("unterminated string");
  }

  ObjP parse_hex_string() {
    ++pos_;
    auto o = mk(Obj::Str);
    int hi = -1;
    while (pos_ < n_) {
      uint8_t c = d_[pos_++];
      if (c == '>') {
        if (hi >= 0) o->s.push_back((char)(hi * 16));
        return o;
      }
      int v = hexval(c);
      if (v < 0) continue;
      if (hi < 0) hi = v;
      else { o->s.push_back((char)(hi * 16 + v)); hi = -1; }
    }
    throw PdfError("unterminated hex string");
  }

  ObjP parse_dict(bool allow_ref, int depth) {
    pos_ += 2;
    auto o = mk(Obj::Dict);
    while (true) {
      skip_ws();
      if (pos_ >= n_) throw PdfError("unterminated dictionary");
      if (d_[pos_] == '>' && pos_ + 1 < n_ && d_[pos_ + 1] == '>') { pos_ += 2; break; }
      if (d_[pos_] != '/') {
        // tolerate garbage: skip one token
        auto junk = parse_object(allow_ref, depth + 1);
        if (!junk) { std::string kw = read_keyword(); if (kw.empty()) ++pos_; }
        continue;
      }
      auto key = parse_name();
      auto val = parse_object(allow_ref, depth + 1);
      if (!val) val = mk(Obj::Null);
      o->dict[key->s] = val;
    }
    return o;
  }

  ObjP parse_number(bool allow_ref) {
    size_t s = pos_;
    bool real = false;
    if (d_[pos_] == '+' || d_[pos_] == '-') ++pos_;
    while (pos_ < n_ && ((d_[pos_] >= '0' && d_[pos_] <= '9') || d_[pos_] == '.')) {
      if (d_[pos_] == '.') real = true;
      ++pos_;
    }
    std::string tok((const char*)d_ + s, pos_ - s);
    if (real) {
      auto o = mk(Obj::Real);
      try { o->r = std::stod(tok); } catch (...) { o->r = 0; }
      return o;
    }
    auto o = mk(Obj::Int);
    try { o->i = std::stoll(tok); } catch (...) { o->i = 0; }
    if (allow_ref && o->i >= 0 && tok[0] != '+' && tok[0] != '-') {
      // look ahead for "<gen> R"
      size_t save = pos_;
      skip_ws();
      size_t g0 = pos_;
      while (pos_ < n_ && d_[pos_] >= '0' && d_[pos_] <= '9') ++pos_;
      if (pos_ > g0 && pos_ - g0 <= 5) {
        long long gen = std::stoll(std::string((const char*)d_ + g0, pos_ - g0));
        skip_ws();
        if (pos_ < n_ && d_[pos_] == 'R' && (pos_ + 1 >= n_ || is_ws(d_[pos_ + 1]) || is_delim(d_[pos_ + 1]))) {
          ++pos_;
          auto r = mk(Obj::Ref);
          r->ref_num = (int)o->i;
          r->ref_gen = (int)gen;
          return r;
        }
      }
      pos_ = save;
    }
    return o;
  }

  const uint8_t* d_;
  size_t n_, pos_;
};

}  // namespace dlpdf
