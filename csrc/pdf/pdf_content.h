// Content-stream interpreter: text operators -> positioned spans -> reading-order text;
// image XObject placements -> page coverage statistics (used by the TXT-vs-OCR classifier).
#pragma once
#include <cstdio>
#include "pdf_fonts.h"

namespace dlpdf {

struct Matrix {
  double a = 1, b = 0, c = 0, d = 1, e = 0, f = 0;
  // this * m  (apply this first, then m) in PDF row-vector convention
  Matrix mul(const Matrix& m) const {
    return {a * m.a + b * m.c, a * m.b + b * m.d, c * m.a + d * m.c, c * m.b + d * m.d,
            e * m.a + f * m.c + m.e, e * m.b + f * m.d + m.f};
  }
};

struct Span {
  double x = 0, y = 0;       // baseline origin in page space
  double x_end = 0;          // estimated end (page space, along x)
  double size = 0;           // effective font size in page units
  std::string text;          // UTF-8
  bool invisible = false;    // text render mode 3 (hidden OCR layer)
};

struct ImagePlacement {
  int obj_num = -1;          // image XObject number (-1: inline image)
  double x0 = 0, y0 = 0, x1 = 0, y1 = 0;  // bounding box in page space
  int width = 0, height = 0;
};

struct PageContent {
  std::vector<Span> spans;
  std::vector<ImagePlacement> images;
  size_t unmapped = 0;       // character codes that could not be decoded
  size_t decoded = 0;
};

class ContentInterpreter {
 public:
  ContentInterpreter(Document& doc, PageContent& out) : doc_(doc), out_(out) {}

  void run_page(const Document::Page& pg) {
    ObjP contents = doc_.dget(pg.dict, "Contents");
    std::string data;
    if (contents && contents->type == Obj::Stream) data = doc_.stream_data(contents);
    else if (contents && contents->type == Obj::Arr)
      for (auto& c : contents->arr) {
        ObjP s = doc_.resolve(c);
        if (s && s->type == Obj::Stream) { data += doc_.stream_data(s); data.push_back('\n'); }
      }
    Matrix base;
    run(data, pg.resources, base, 0);
  }

 private:
  struct GState {
    Matrix ctm;
    std::shared_ptr<Font> font;
    double fs = 0, tc = 0, tw = 0, tz = 1, tl = 0, rise = 0;
    int render = 0;
  };

  std::shared_ptr<Font> get_font(const ObjP& resources, const std::string& name) {
    ObjP fonts = doc_.dget(resources, "Font");
    ObjP fd = fonts && fonts->type == Obj::Dict ? fonts->get(name) : nullptr;
    int key = fd && fd->type == Obj::Ref ? fd->ref_num : -1;
    if (key >= 0) {
      auto it = font_cache_.find(key);
      if (it != font_cache_.end()) return it->second;
    }
    auto f = load_font(doc_, doc_.resolve(fd));
    if (key >= 0) font_cache_[key] = f;
    return f;
  }

  void show(GState& gs, Matrix& tm, const std::string& s) {
    if (!gs.font) { gs.font = std::make_shared<Font>(); fill_base_encoding(*gs.font, "WinAnsiEncoding"); }
    const Font& f = *gs.font;
    Matrix trm = Matrix{gs.fs * gs.tz, 0, 0, gs.fs, 0, gs.rise}.mul(tm).mul(gs.ctm);
    Span sp;
    sp.x = trm.e; sp.y = trm.f;
    sp.size = std::sqrt(std::fabs(trm.a * trm.d - trm.b * trm.c));
    sp.invisible = gs.render == 3;
    double adv_total = 0;
    size_t i = 0;
    const int cb = f.composite ? f.code_bytes : 1;
    while (i < s.size()) {
      uint32_t code = (uint8_t)s[i];
      if (cb == 2 && i + 1 < s.size()) code = (code << 8) | (uint8_t)s[i + 1];
      i += (cb == 2 && i + 1 < s.size()) ? 2 : 1;
      if (f.decode(code, sp.text)) ++out_.decoded;
      else { ++out_.unmapped; sp.text += "\xEF\xBF\xBD"; }
      double w = f.width(code) / 1000.0;
      double adv = (w * gs.fs + gs.tc + ((cb == 1 && code == 32) ? gs.tw : 0.0)) * gs.tz;
      adv_total += adv;
    }
    // advance the text matrix
    Matrix t{1, 0, 0, 1, adv_total, 0};
    Matrix tm2 = t.mul(tm);
    Matrix end = Matrix{gs.fs * gs.tz, 0, 0, gs.fs, 0, gs.rise}.mul(tm2).mul(gs.ctm);
    sp.x_end = end.e;
    tm = tm2;
    if (!sp.text.empty()) out_.spans.push_back(std::move(sp));
  }

  void place_image(const GState& gs, int obj_num, int w, int h) {
    // unit square through the CTM
    double xs[4], ys[4];
    const double ux[4] = {0, 1, 0, 1}, uy[4] = {0, 0, 1, 1};
    for (int k = 0; k < 4; ++k) {
      xs[k] = ux[k] * gs.ctm.a + uy[k] * gs.ctm.c + gs.ctm.e;
      ys[k] = ux[k] * gs.ctm.b + uy[k] * gs.ctm.d + gs.ctm.f;
    }
    ImagePlacement ip;
    ip.obj_num = obj_num;
    ip.x0 = *std::min_element(xs, xs + 4); ip.x1 = *std::max_element(xs, xs + 4);
    ip.y0 = *std::min_element(ys, ys + 4); ip.y1 = *std::max_element(ys, ys + 4);
    ip.width = w; ip.height = h;
    out_.images.push_back(ip);
  }

  static double num(const std::vector<ObjP>& st, size_t idx_from_end) {
    if (st.size() <= idx_from_end) return 0;
    const ObjP& o = st[st.size() - 1 - idx_from_end];
    return o && o->is_num() ? o->num() : 0;
  }

  void run(const std::string& data, const ObjP& resources, const Matrix& base, int depth) {
    if (depth > 8) return;
    Lexer lx((const uint8_t*)data.data(), data.size());
    std::vector<ObjP> st;
    std::vector<GState> stack;
    GState gs;
    gs.ctm = base;
    Matrix tm, tlm;
    size_t guard = 0;
    while (!lx.eof()) {
      if (++guard > 50000000) break;
      ObjP o;
      try { o = lx.parse_object(false); } catch (const PdfError&) { break; }
      if (o) { if (st.size() < 4096) st.push_back(o); continue; }
      std::string op = lx.read_keyword();
      if (op.empty()) { lx.seek(lx.pos() + 1); continue; }
      if (op == "q") { if (stack.size() < 256) stack.push_back(gs); }
      else if (op == "Q") { if (!stack.empty()) { gs = stack.back(); stack.pop_back(); } }
      else if (op == "cm") {
        Matrix m{num(st, 5), num(st, 4), num(st, 3), num(st, 2), num(st, 1), num(st, 0)};
        gs.ctm = m.mul(gs.ctm);
      } else if (op == "BT") { tm = Matrix(); tlm = Matrix(); }
      else if (op == "Tf") {
        gs.fs = num(st, 0);
        if (st.size() >= 2 && st[st.size() - 2]->type == Obj::Name) gs.font = get_font(resources, st[st.size() - 2]->s);
      } else if (op == "Td" || op == "TD") {
        double tx = num(st, 1), ty = num(st, 0);
        if (op == "TD") gs.tl = -ty;
        tlm = Matrix{1, 0, 0, 1, tx, ty}.mul(tlm);
        tm = tlm;
      } else if (op == "Tm") {
        tlm = Matrix{num(st, 5), num(st, 4), num(st, 3), num(st, 2), num(st, 1), num(st, 0)};
        tm = tlm;
      } else if (op == "T*") { tlm = Matrix{1, 0, 0, 1, 0, -gs.tl}.mul(tlm); tm = tlm; }
      else if (op == "Tc") gs.tc = num(st, 0);
      else if (op == "Tw") gs.tw = num(st, 0);
      else if (op == "Tz") gs.tz = num(st, 0) / 100.0;
      else if (op == "TL") gs.tl = num(st, 0);
      else if (op == "Ts") gs.rise = num(st, 0);
      else if (op == "Tr") gs.render = (int)num(st, 0);
      else if (op == "Tj") { if (!st.empty() && st.back()->type == Obj::Str) show(gs, tm, st.back()->s); }
      else if (op == "'") {
        tlm = Matrix{1, 0, 0, 1, 0, -gs.tl}.mul(tlm); tm = tlm;
        if (!st.empty() && st.back()->type == Obj::Str) show(gs, tm, st.back()->s);
      } else if (op == "\"") {
        gs.tw = num(st, 2); gs.tc = num(st, 1);
        tlm = Matrix{1, 0, 0, 1, 0, -gs.tl}.mul(tlm); tm = tlm;
        if (!st.empty() && st.back()->type == Obj::Str) show(gs, tm, st.back()->s);
      } else if (op == "TJ") {
        if (!st.empty() && st.back()->type == Obj::Arr) {
          for (auto& e : st.back()->arr) {
            if (e->type == Obj::Str) show(gs, tm, e->s);
            else if (e->is_num()) {
              double tx = -e->num() / 1000.0 * gs.fs * gs.tz;
              tm = Matrix{1, 0, 0, 1, tx, 0}.mul(tm);
            }
          }
        }
      } else if (op == "Do") {
        if (!st.empty() && st.back()->type == Obj::Name) {
          ObjP xo = doc_.dget(resources, "XObject");
          ObjP ref = xo && xo->type == Obj::Dict ? xo->get(st.back()->s) : nullptr;
          ObjP x = doc_.resolve(ref);
          if (x && x->type == Obj::Stream) {
            ObjP sub = doc_.dget(x, "Subtype");
            if (sub && sub->type == Obj::Name && sub->s == "Image") {
              place_image(gs, ref && ref->type == Obj::Ref ? ref->ref_num : -1, (int)to_int(doc_.dget(x, "Width")),
                          (int)to_int(doc_.dget(x, "Height")));
            } else if (sub && sub->type == Obj::Name && sub->s == "Form") {
              Matrix fm;
              ObjP m = doc_.dget(x, "Matrix");
              if (m && m->type == Obj::Arr && m->arr.size() >= 6) {
                double v[6];
                for (int k = 0; k < 6; ++k) { ObjP e = doc_.resolve(m->arr[k]); v[k] = e && e->is_num() ? e->num() : 0; }
                fm = Matrix{v[0], v[1], v[2], v[3], v[4], v[5]};
              }
              ObjP fres = doc_.dget(x, "Resources");
              run(doc_.stream_data(x), (fres && fres->type == Obj::Dict) ? fres : resources, fm.mul(gs.ctm), depth + 1);
            }
          }
        }
      } else if (op == "BI") {
        // inline image: parse the key/value pairs up to ID, then skip the binary data to EI
        ObjP wv, hv;
        while (!lx.eof()) {
          if (lx.peek_keyword("ID")) { lx.read_keyword(); break; }
          ObjP k;
          try { k = lx.parse_object(false); } catch (const PdfError&) { break; }
          if (!k) { std::string junk = lx.read_keyword(); if (junk.empty()) lx.seek(lx.pos() + 1); continue; }
          ObjP v;
          try { v = lx.parse_object(false); } catch (const PdfError&) { break; }
          if (k->type == Obj::Name && v) {
            if (k->s == "W" || k->s == "Width") wv = v;
            if (k->s == "H" || k->s == "Height") hv = v;
          }
        }
        size_t p = lx.pos() + 1;
        const uint8_t* d = lx.data();
        size_t n = lx.size();
        while (p + 2 < n && !(is_ws(d[p]) && d[p + 1] == 'E' && d[p + 2] == 'I' && (p + 3 >= n || is_ws(d[p + 3])))) ++p;
        lx.seek(std::min(n, p + 3));
        place_image(gs, -1, (int)to_int(wv), (int)to_int(hv));
      }
      st.clear();
    }
  }

  Document& doc_;
  PageContent& out_;
  std::unordered_map<int, std::shared_ptr<Font>> font_cache_;
};

// Spans -> paragraphs.  Lines are formed by baseline proximity, ordered top-to-bottom
// (PDF y grows upward), spans inside a line left-to-right with a space inserted across
// visible gaps; lines are merged into a paragraph until a vertical gap larger than
// 1.6x the line pitch (or a clear font-size change) appears.  Paragraphs are separated
// by a blank line ("\n\n") - each one becomes a 'text' item of the content list.
inline std::string assemble_text(std::vector<Span> spans, bool include_invisible) {
  std::vector<Span*> v;
  for (auto& s : spans)
    if ((include_invisible || !s.invisible) && !s.text.empty()) v.push_back(&s);
  if (v.empty()) return std::string();
  std::stable_sort(v.begin(), v.end(), [](const Span* a, const Span* b) {
    if (std::fabs(a->y - b->y) > 0.5 * std::max(1.0, std::min(a->size, b->size))) return a->y > b->y;
    return a->x < b->x;
  });
  struct Line { double y, size; std::string text; };
  std::vector<Line> lines;
  const Span* prev = nullptr;
  for (const Span* s : v) {
    double sz = std::max(1.0, s->size);
    if (!prev || std::fabs(prev->y - s->y) > 0.5 * std::max(1.0, std::min(prev->size, s->size))) {
      lines.push_back({s->y, sz, s->text});
    } else {
      Line& L = lines.back();
      double gap = s->x - prev->x_end;
      bool need_space = gap > 0.2 * sz && !L.text.empty() && L.text.back() != ' ' && s->text[0] != ' ';
      if (need_space) L.text.push_back(' ');
      L.text += s->text;
      L.size = std::max(L.size, sz);
    }
    prev = s;
  }
  std::string out;
  for (size_t i = 0; i < lines.size(); ++i) {
    std::string t = lines[i].text;
    while (!t.empty() && (t.back() == ' ' || t.back() == '\t')) t.pop_back();
    size_t b = 0;
    while (b < t.size() && t[b] == ' ') ++b;
    t = t.substr(b);
    if (t.empty()) continue;
    if (!out.empty()) {
      double pitch = lines[i - 1].y - lines[i].y;
      double ref = std::max(lines[i].size, lines[i - 1].size);
      bool para = pitch > 1.6 * ref * 1.25 || pitch < 0 || std::fabs(lines[i].size - lines[i - 1].size) > 0.25 * ref;
      out += para ? "\n\n" : "\n";
    }
    out += t;
  }
  return out;
}

// Spans -> positioned text lines for layout analysis (multi-column reading order).  Spans on one baseline form a
// line; a horizontal gap wider than 2.5 font sizes (a column gutter, a table cell boundary) starts a new line, so the
// two columns of a page never fuse into one record.  Output: one record per line,
//   x0 \t y0 \t x1 \t y1 \t size \t text \n        (page space, y up; tabs / newlines inside text become spaces)
// in top-to-bottom, left-to-right order.  Ordering across columns is the caller's job.
inline std::string assemble_lines(std::vector<Span> spans, bool include_invisible) {
  std::vector<Span*> v;
  for (auto& s : spans)
    if ((include_invisible || !s.invisible) && !s.text.empty()) v.push_back(&s);
  if (v.empty()) return std::string();
  std::stable_sort(v.begin(), v.end(), [](const Span* a, const Span* b) {
    if (std::fabs(a->y - b->y) > 0.5 * std::max(1.0, std::min(a->size, b->size))) return a->y > b->y;
    return a->x < b->x;
  });
  struct Line { double x0, x1, y, size; std::string text; };
  std::vector<Line> lines;
  const Span* prev = nullptr;
  for (const Span* s : v) {
    const double sz = std::max(1.0, s->size);
    const bool same_base = prev && std::fabs(prev->y - s->y) <= 0.5 * std::max(1.0, std::min(prev->size, s->size));
    const double gap = prev ? s->x - prev->x_end : 0.0;
    if (!same_base || gap > 2.5 * sz || gap < -4.0 * sz) {
      lines.push_back({s->x, std::max(s->x, s->x_end), s->y, sz, s->text});
    } else {
      Line& L = lines.back();
      if (gap > 0.2 * sz && !L.text.empty() && L.text.back() != ' ' && s->text[0] != ' ') L.text.push_back(' ');
      L.text += s->text;
      L.size = std::max(L.size, sz);
      L.x1 = std::max(L.x1, s->x_end);
    }
    prev = s;
  }
  std::string out;
  char num[160];
  for (auto& L : lines) {
    std::string t = L.text;
    for (auto& c : t) if (c == '\t' || c == '\n' || c == '\r') c = ' ';
    while (!t.empty() && t.back() == ' ') t.pop_back();
    size_t b = 0;
    while (b < t.size() && t[b] == ' ') ++b;
    t = t.substr(b);
    if (t.empty()) continue;
    snprintf(num, sizeof num, "%.2f\t%.2f\t%.2f\t%.2f\t%.2f\t", L.x0, L.y - 0.25 * L.size, L.x1, L.y + 0.85 * L.size, L.size);
    out += num;
    out += t;
    out.push_back('\n');
  }
  return out;
}

}  // namespace dlpdf
