// Font decoding for text extraction: simple-font encodings (+/Differences), ToUnicode CMaps,
// composite (Type0) fonts with 2-byte codes, and glyph widths for text positioning.
#pragma once
#include <array>

#include "pdf_document.h"

namespace dlpdf {

inline void append_utf8(std::string& out, uint32_t cp) {
  if (cp < 0x80) out.push_back((char)cp);
  else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) {
    out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
    out.push_back((char)(0x80 | (cp & 0x3F)));
  } else {
    out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
    out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
  }
}

// cp1252 0x80..0x9F
static const uint16_t kWinAnsiHigh[32] = {0x20AC, 0x2022, 0x201A, 0x0192, 0x201E, 0x2026, 0x2020, 0x2021, 0x02C6, 0x2030, 0x0160,
                                          0x2039, 0x0152, 0x2022, 0x017D, 0x2022, 0x2022, 0x2018, 0x2019, 0x201C, 0x201D, 0x2022,
                                          0x2013, 0x2014, 0x02DC, 0x2122, 0x0161, 0x203A, 0x0153, 0x2022, 0x017E, 0x0178};
// MacRoman 0x80..0xFF
static const uint16_t kMacRomanHigh[128] = {
    0x00C4, 0x00C5, 0x00C7, 0x00C9, 0x00D1, 0x00D6, 0x00DC, 0x00E1, 0x00E0, 0x00E2, 0x00E4, 0x00E3, 0x00E5, 0x00E7, 0x00E9, 0x00E8,
    0x00EA, 0x00EB, 0x00ED, 0x00EC, 0x00EE, 0x00EF, 0x00F1, 0x00F3, 0x00F2, 0x00F4, 0x00F6, 0x00F5, 0x00FA, 0x00F9, 0x00FB, 0x00FC,
    0x2020, 0x00B0, 0x00A2, 0x00A3, 0x00A7, 0x2022, 0x00B6, 0x00DF, 0x00AE, 0x00A9, 0x2122, 0x00B4, 0x00A8, 0x2260, 0x00C6, 0x00D8,
    0x221E, 0x00B1, 0x2264, 0x2265, 0x00A5, 0x00B5, 0x2202, 0x2211, 0x220F, 0x03C0, 0x222B, 0x00AA, 0x00BA, 0x03A9, 0x00E6, 0x00F8,
    0x00BF, 0x00A1, 0x00AC, 0x221A, 0x0192, 0x2248, 0x2206, 0x00AB, 0x00BB, 0x2026, 0x00A0, 0x00C0, 0x00C3, 0x00D5, 0x0152, 0x0153,
    0x2013, 0x2014, 0x201C, 0x201D, 0x2018, 0x2019, 0x00F7, 0x25CA, 0x00FF, 0x0178, 0x2044, 0x20AC, 0x2039, 0x203A, 0xFB01, 0xFB02,
    0x2021, 0x00B7, 0x201A, 0x201E, 0x2030, 0x00C2, 0x00CA, 0x00C1, 0x00CB, 0x00C8, 0x00CD, 0x00CE, 0x00CF, 0x00CC, 0x00D3, 0x00D4,
    0xF8FF, 0x00D2, 0x00DA, 0x00DB, 0x00D9, 0x0131, 0x02C6, 0x02DC, 0x00AF, 0x02D8, 0x02D9, 0x02DA, 0x00B8, 0x02DD, 0x02DB, 0x02C7};

struct GlyphName { const char* name; uint16_t cp; };
// Subset of the Adobe Glyph List covering Latin text, punctuation and ligatures
// (names of the form uniXXXX / uXXXX are handled algorithmically).
static const GlyphName kGlyphs[] = {
    {"space", 0x20}, {"exclam", 0x21}, {"quotedbl", 0x22}, {"numbersign", 0x23}, {"dollar", 0x24}, {"percent", 0x25},
    {"ampersand", 0x26}, {"quotesingle", 0x27}, {"parenleft", 0x28}, {"parenright", 0x29}, {"asterisk", 0x2A}, {"plus", 0x2B},
    {"comma", 0x2C}, {"hyphen", 0x2D}, {"period", 0x2E}, {"slash", 0x2F}, {"zero", 0x30}, {"one", 0x31}, {"two", 0x32},
    {"three", 0x33}, {"four", 0x34}, {"five", 0x35}, {"six", 0x36}, {"seven", 0x37}, {"eight", 0x38}, {"nine", 0x39},
    {"colon", 0x3A}, {"semicolon", 0x3B}, {"less", 0x3C}, {"equal", 0x3D}, {"greater", 0x3E}, {"question", 0x3F}, {"at", 0x40},
    {"bracketleft", 0x5B}, {"backslash", 0x5C}, {"bracketright", 0x5D}, {"asciicircum", 0x5E}, {"underscore", 0x5F},
    {"grave", 0x60}, {"braceleft", 0x7B}, {"bar", 0x7C}, {"braceright", 0x7D}, {"asciitilde", 0x7E},
    {"exclamdown", 0xA1}, {"cent", 0xA2}, {"sterling", 0xA3}, {"currency", 0xA4}, {"yen", 0xA5}, {"brokenbar", 0xA6},
    {"section", 0xA7}, {"dieresis", 0xA8}, {"copyright", 0xA9}, {"ordfeminine", 0xAA}, {"guillemotleft", 0xAB},
    {"logicalnot", 0xAC}, {"registered", 0xAE}, {"macron", 0xAF}, {"degree", 0xB0}, {"plusminus", 0xB1}, {"twosuperior", 0xB2},
    {"threesuperior", 0xB3}, {"acute", 0xB4}, {"mu", 0xB5}, {"paragraph", 0xB6}, {"periodcentered", 0xB7}, {"cedilla", 0xB8},
    {"onesuperior", 0xB9}, {"ordmasculine", 0xBA}, {"guillemotright", 0xBB}, {"onequarter", 0xBC}, {"onehalf", 0xBD},
    {"threequarters", 0xBE}, {"questiondown", 0xBF}, {"Agrave", 0xC0}, {"Aacute", 0xC1}, {"Acircumflex", 0xC2},
    {"Atilde", 0xC3}, {"Adieresis", 0xC4}, {"Aring", 0xC5}, {"AE", 0xC6}, {"Ccedilla", 0xC7}, {"Egrave", 0xC8},
    {"Eacute", 0xC9}, {"Ecircumflex", 0xCA}, {"Edieresis", 0xCB}, {"Igrave", 0xCC}, {"Iacute", 0xCD}, {"Icircumflex", 0xCE},
    {"Idieresis", 0xCF}, {"Eth", 0xD0}, {"Ntilde", 0xD1}, {"Ograve", 0xD2}, {"Oacute", 0xD3}, {"Ocircumflex", 0xD4},
    {"Otilde", 0xD5}, {"Odieresis", 0xD6}, {"multiply", 0xD7}, {"Oslash", 0xD8}, {"Ugrave", 0xD9}, {"Uacute", 0xDA},
    {"Ucircumflex", 0xDB}, {"Udieresis", 0xDC}, {"Yacute", 0xDD}, {"Thorn", 0xDE}, {"germandbls", 0xDF}, {"agrave", 0xE0},
    {"aacute", 0xE1}, {"acircumflex", 0xE2}, {"atilde", 0xE3}, {"adieresis", 0xE4}, {"aring", 0xE5}, {"ae", 0xE6},
    {"ccedilla", 0xE7}, {"egrave", 0xE8}, {"eacute", 0xE9}, {"ecircumflex", 0xEA}, {"edieresis", 0xEB}, {"igrave", 0xEC},
    {"iacute", 0xED}, {"icircumflex", 0xEE}, {"idieresis", 0xEF}, {"eth", 0xF0}, {"ntilde", 0xF1}, {"ograve", 0xF2},
    {"oacute", 0xF3}, {"ocircumflex", 0xF4}, {"otilde", 0xF5}, {"odieresis", 0xF6}, {"divide", 0xF7}, {"oslash", 0xF8},
    {"ugrave", 0xF9}, {"uacute", 0xFA}, {"ucircumflex", 0xFB}, {"udieresis", 0xFC}, {"yacute", 0xFD}, {"thorn", 0xFE},
    {"ydieresis", 0xFF}, {"OE", 0x152}, {"oe", 0x153}, {"Scaron", 0x160}, {"scaron", 0x161}, {"Ydieresis", 0x178},
    {"Zcaron", 0x17D}, {"zcaron", 0x17E}, {"florin", 0x192}, {"circumflex", 0x2C6}, {"caron", 0x2C7}, {"breve", 0x2D8},
    {"dotaccent", 0x2D9}, {"ring", 0x2DA}, {"ogonek", 0x2DB}, {"tilde", 0x2DC}, {"hungarumlaut", 0x2DD}, {"dotlessi", 0x131},
    {"Lslash", 0x141}, {"lslash", 0x142}, {"endash", 0x2013}, {"emdash", 0x2014}, {"quoteleft", 0x2018},
    {"quoteright", 0x2019}, {"quotesinglbase", 0x201A}, {"quotedblleft", 0x201C}, {"quotedblright", 0x201D},
    {"quotedblbase", 0x201E}, {"dagger", 0x2020}, {"daggerdbl", 0x2021}, {"bullet", 0x2022}, {"ellipsis", 0x2026},
    {"perthousand", 0x2030}, {"guilsinglleft", 0x2039}, {"guilsinglright", 0x203A}, {"fraction", 0x2044}, {"Euro", 0x20AC},
    {"trademark", 0x2122}, {"minus", 0x2212}, {"fi", 0xFB01}, {"fl", 0xFB02}, {"ff", 0xFB00}, {"ffi", 0xFB03},
    {"ffl", 0xFB04}, {"nbspace", 0xA0}, {"sfthyphen", 0xAD}, {"Delta", 0x394}, {"Omega", 0x3A9}, {"pi", 0x3C0},
    {"alpha", 0x3B1}, {"beta", 0x3B2}, {"gamma", 0x3B3}, {"lessequal", 0x2264}, {"greaterequal", 0x2265},
    {"notequal", 0x2260}, {"infinity", 0x221E}, {"approxequal", 0x2248}, {"arrowright", 0x2192}, {"arrowleft", 0x2190},
};

inline uint32_t glyph_to_unicode(const std::string& name) {
  if (name.size() == 1) return (uint8_t)name[0];  // A-Z a-z
  static std::unordered_map<std::string, uint16_t>* tbl = nullptr;
  if (!tbl) {
    tbl = new std::unordered_map<std::string, uint16_t>();
    for (const auto& g : kGlyphs) (*tbl)[g.name] = g.cp;
  }
  auto it = tbl->find(name);
  if (it != tbl->end()) return it->second;
  auto hex = [&](size_t from, size_t len) -> long {
    long v = 0;
    if (from + len > name.size()) return -1;
    for (size_t i = from; i < from + len; ++i) { int h = hexval((uint8_t)name[i]); if (h < 0) return -1; v = v * 16 + h; }
    return v;
  };
  if (name.size() >= 7 && name.compare(0, 3, "uni") == 0) { long v = hex(3, 4); if (v >= 0) return (uint32_t)v; }
  if (name.size() >= 5 && name[0] == 'u') { long v = hex(1, name.size() - 1); if (v > 0 && v < 0x110000) return (uint32_t)v; }
  // "name.suffix" (e.g. a.sc): strip the suffix
  size_t dot = name.find('.');
  if (dot != std::string::npos && dot > 0) return glyph_to_unicode(name.substr(0, dot));
  return 0;
}

struct Font {
  bool composite = false;          // Type0: 2-byte codes
  bool has_tounicode = false;
  std::array<uint32_t, 256> enc{};  // simple-font code -> unicode (0 = unmapped)
  std::unordered_map<uint32_t, std::u32string> tounicode;
  struct Range { uint32_t lo, hi, base; };
  std::vector<Range> bfranges;
  int code_bytes = 1;
  // widths in 1/1000 em
  int first_char = 0;
  std::vector<float> widths;
  float default_width = 500.f;
  std::unordered_map<uint32_t, float> cid_widths;

  float width(uint32_t code) const {
    if (!composite) {
      long idx = (long)code - first_char;
      if (idx >= 0 && (size_t)idx < widths.size() && widths[idx] > 0) return widths[idx];
      return default_width;
    }
    auto it = cid_widths.find(code);
    return it == cid_widths.end() ? default_width : it->second;
  }

  // Decode one character code to UTF-8; returns false when the code is unmappable.
  bool decode(uint32_t code, std::string& out) const {
    if (has_tounicode) {
      auto it = tounicode.find(code);
      if (it != tounicode.end()) { for (char32_t c : it->second) append_utf8(out, c); return true; }
      for (const auto& r : bfranges)
        if (code >= r.lo && code <= r.hi) { append_utf8(out, r.base + (code - r.lo)); return true; }
    }
    if (!composite) {
      uint32_t u = code < 256 ? enc[code] : 0;
      if (u) { append_utf8(out, u); return true; }
      return false;
    }
    // Identity-H without ToUnicode: the code is a glyph id, not recoverable
    return false;
  }
};

inline void fill_base_encoding(Font& f, const std::string& name) {
  for (int c = 0; c < 256; ++c) f.enc[c] = 0;
  for (int c = 32; c < 127; ++c) f.enc[c] = c;
  if (name == "MacRomanEncoding") {
    for (int c = 0; c < 128; ++c) f.enc[128 + c] = kMacRomanHigh[c];
  } else if (name == "StandardEncoding") {
    f.enc[0x27] = 0x2019; f.enc[0x60] = 0x2018;
    static const struct { uint8_t c; uint16_t u; } hi[] = {
        {0xA1, 0xA1}, {0xA2, 0xA2}, {0xA3, 0xA3}, {0xA4, 0x2044}, {0xA5, 0xA5}, {0xA6, 0x192}, {0xA7, 0xA7}, {0xA8, 0xA4},
        {0xA9, 0x27}, {0xAA, 0x201C}, {0xAB, 0xAB}, {0xAC, 0x2039}, {0xAD, 0x203A}, {0xAE, 0xFB01}, {0xAF, 0xFB02},
        {0xB1, 0x2013}, {0xB2, 0x2020}, {0xB3, 0x2021}, {0xB4, 0xB7}, {0xB6, 0xB6}, {0xB7, 0x2022}, {0xB8, 0x201A},
        {0xB9, 0x201E}, {0xBA, 0x201D}, {0xBB, 0xBB}, {0xBC, 0x2026}, {0xBD, 0x2030}, {0xBF, 0xBF}, {0xC1, 0x60}, {0xC2, 0xB4},
        {0xC3, 0x2C6}, {0xC4, 0x2DC}, {0xC5, 0xAF}, {0xC6, 0x2D8}, {0xC7, 0x2D9}, {0xC8, 0xA8}, {0xCA, 0x2DA}, {0xCB, 0xB8},
        {0xCD, 0x2DD}, {0xCE, 0x2DB}, {0xCF, 0x2C7}, {0xD0, 0x2014}, {0xE1, 0xC6}, {0xE3, 0xAA}, {0xE8, 0x141}, {0xE9, 0xD8},
        {0xEA, 0x152}, {0xEB, 0xBA}, {0xF1, 0xE6}, {0xF5, 0x131}, {0xF8, 0x142}, {0xF9, 0xF8}, {0xFA, 0x153}, {0xFB, 0xDF}};
    for (const auto& e : hi) f.enc[e.c] = e.u;
  } else {
    // WinAnsiEncoding (also the fallback for unknown / missing encodings: cp1252 is a superset of Latin-1 text)
    for (int c = 0; c < 32; ++c) f.enc[128 + c] = kWinAnsiHigh[c];
    for (int c = 160; c < 256; ++c) f.enc[c] = c;
    f.enc[0xAD] = 0x2D;
  }
}

// Parse the bfchar / bfrange sections of a ToUnicode CMap.
inline void parse_tounicode(Font& f, const std::string& cmap) {
  Lexer lx((const uint8_t*)cmap.data(), cmap.size());
  auto code_of = [](const std::string& s) { uint32_t v = 0; for (unsigned char c : s) v = (v << 8) | c; return v; };
  auto utf16_of = [](const std::string& s) {
    std::u32string out;
    for (size_t i = 0; i + 1 < s.size(); i += 2) {
      uint32_t u = ((uint8_t)s[i] << 8) | (uint8_t)s[i + 1];
      if (u >= 0xD800 && u < 0xDC00 && i + 3 < s.size()) {
        uint32_t lo = ((uint8_t)s[i + 2] << 8) | (uint8_t)s[i + 3];
        u = 0x10000 + ((u - 0xD800) << 10) + (lo - 0xDC00);
        i += 2;
      }
      out.push_back(u);
    }
    if (s.size() == 1) out.push_back((uint8_t)s[0]);
    return out;
  };
  std::vector<ObjP> operands;
  int maxlen = 1;
  while (!lx.eof()) {
    ObjP o;
    try { o = lx.parse_object(false); } catch (const PdfError&) { break; }
    if (o) { operands.push_back(o); continue; }
    std::string kw = lx.read_keyword();
    if (kw.empty()) { lx.seek(lx.pos() + 1); continue; }
    if (kw == "endbfchar") {
      for (size_t i = 0; i + 1 < operands.size(); i += 2) {
        if (operands[i]->type != Obj::Str) continue;
        maxlen = std::max<int>(maxlen, (int)operands[i]->s.size());
        if (operands[i + 1]->type == Obj::Str) f.tounicode[code_of(operands[i]->s)] = utf16_of(operands[i + 1]->s);
      }
    } else if (kw == "endbfrange") {
      for (size_t i = 0; i + 2 < operands.size(); i += 3) {
        if (operands[i]->type != Obj::Str || operands[i + 1]->type != Obj::Str) continue;
        maxlen = std::max<int>(maxlen, (int)operands[i]->s.size());
        uint32_t lo = code_of(operands[i]->s), hi = code_of(operands[i + 1]->s);
        if (hi < lo || hi - lo > 65535) continue;
        if (operands[i + 2]->type == Obj::Str) {
          std::u32string b = utf16_of(operands[i + 2]->s);
          if (b.size() == 1) f.bfranges.push_back({lo, hi, (uint32_t)b[0]});
          else if (!b.empty())
            for (uint32_t c = lo; c <= hi; ++c) { std::u32string v = b; v.back() += (c - lo); f.tounicode[c] = v; }
        } else if (operands[i + 2]->type == Obj::Arr) {
          for (uint32_t c = lo; c <= hi && (c - lo) < operands[i + 2]->arr.size(); ++c)
            if (operands[i + 2]->arr[c - lo]->type == Obj::Str) f.tounicode[c] = utf16_of(operands[i + 2]->arr[c - lo]->s);
        }
      }
    } else if (kw == "endcodespacerange") {
      for (auto& op : operands) if (op->type == Obj::Str) maxlen = std::max<int>(maxlen, (int)op->s.size());
    }
    if (kw.compare(0, 5, "begin") != 0) operands.clear();
    else operands.clear();
  }
  f.has_tounicode = !f.tounicode.empty() || !f.bfranges.empty();
  if (f.composite) f.code_bytes = std::max(f.code_bytes, std::min(maxlen, 2));
}

inline std::shared_ptr<Font> load_font(Document& doc, const ObjP& fd) {
  auto f = std::make_shared<Font>();
  fill_base_encoding(*f, "WinAnsiEncoding");
  if (!fd || fd->type != Obj::Dict) return f;
  ObjP sub = doc.dget(fd, "Subtype");
  f->composite = sub && sub->type == Obj::Name && sub->s == "Type0";
  if (f->composite) {
    f->code_bytes = 2;
    f->default_width = 1000.f;
    ObjP desc = doc.dget(fd, "DescendantFonts");
    if (desc && desc->type == Obj::Arr && !desc->arr.empty()) {
      ObjP cf = doc.resolve(desc->arr[0]);
      if (cf && cf->type == Obj::Dict) {
        ObjP dw = doc.dget(cf, "DW");
        if (dw && dw->is_num()) f->default_width = (float)dw->num();
        ObjP w = doc.dget(cf, "W");
        if (w && w->type == Obj::Arr) {
          size_t i = 0;
          while (i < w->arr.size()) {
            ObjP a = doc.resolve(w->arr[i]);
            if (!a || !a->is_num() || i + 1 >= w->arr.size()) break;
            ObjP b = doc.resolve(w->arr[i + 1]);
            if (b && b->type == Obj::Arr) {
              uint32_t c = (uint32_t)a->num();
              for (auto& e : b->arr) { ObjP ev = doc.resolve(e); if (ev && ev->is_num()) f->cid_widths[c] = (float)ev->num(); ++c; }
              i += 2;
            } else if (b && b->is_num() && i + 2 < w->arr.size()) {
              ObjP cw = doc.resolve(w->arr[i + 2]);
              uint32_t lo = (uint32_t)a->num(), hi = (uint32_t)b->num();
              if (cw && cw->is_num() && hi >= lo && hi - lo < 65536)
                for (uint32_t c = lo; c <= hi; ++c) f->cid_widths[c] = (float)cw->num();
              i += 3;
            } else break;
          }
        }
      }
    }
    ObjP enc = doc.dget(fd, "Encoding");
    if (enc && enc->type == Obj::Name && enc->s.find("-H") == std::string::npos && enc->s.find("-V") == std::string::npos)
      f->code_bytes = 1;
  } else {
    ObjP enc = doc.dget(fd, "Encoding");
    ObjP base = doc.dget(fd, "BaseFont");
    std::string bn = base && base->type == Obj::Name ? base->s : "";
    bool symbolic = bn.find("Symbol") != std::string::npos || bn.find("ZapfDingbats") != std::string::npos;
    if (enc && enc->type == Obj::Name) fill_base_encoding(*f, enc->s);
    else if (enc && enc->type == Obj::Dict) {
      ObjP be = doc.dget(enc, "BaseEncoding");
      fill_base_encoding(*f, be && be->type == Obj::Name ? be->s : (symbolic ? "" : "StandardEncoding"));
      ObjP diff = doc.dget(enc, "Differences");
      if (diff && diff->type == Obj::Arr) {
        int code = 0;
        for (auto& e : diff->arr) {
          ObjP ev = doc.resolve(e);
          if (!ev) continue;
          if (ev->is_num()) code = (int)ev->num();
          else if (ev->type == Obj::Name) {
            if (code >= 0 && code < 256) { uint32_t u = glyph_to_unicode(ev->s); if (u) f->enc[code] = u; }
            ++code;
          }
        }
      }
    }
    ObjP fc = doc.dget(fd, "FirstChar");
    ObjP ws = doc.dget(fd, "Widths");
    if (fc && fc->is_num()) f->first_char = (int)fc->num();
    if (ws && ws->type == Obj::Arr)
      for (auto& e : ws->arr) { ObjP ev = doc.resolve(e); f->widths.push_back(ev && ev->is_num() ? (float)ev->num() : 0.f); }
    if (bn.find("Courier") != std::string::npos) f->default_width = 600.f;
    else f->default_width = 520.f;  // Helvetica / Times average advance
    ObjP fdesc = doc.dget(fd, "FontDescriptor");
    if (fdesc && fdesc->type == Obj::Dict) {
      ObjP mw = doc.dget(fdesc, "MissingWidth");
      if (mw && mw->is_num() && mw->num() > 0) f->default_width = (float)mw->num();
    }
  }
  ObjP tu = doc.dget(fd, "ToUnicode");
  if (tu && tu->type == Obj::Stream) {
    try { parse_tounicode(*f, doc.stream_data(tu)); } catch (const PdfError&) {}
  }
  return f;
}

}  // namespace dlpdf
