// C ABI of the native PDF reader (loaded from Python with ctypes; every call releases the GIL,
// so page analysis and image decoding run in parallel on the host thread pool).
//
// Replaces PyMuPDF + pdfminer in the reference's stack:
//   services/ocr_service/ocr_service.py:210  PymuDocDataset(pdf_bytes)
//   services/ocr_service/ocr_service.py:213  ds.classify()
#include <algorithm>
#include <mutex>

#include "ccitt_fax.h"
#include "jpeg_scan.h"
#include "jpeg_sync_host.h"
#include "pdf_content.h"

using namespace dlpdf;

namespace {

struct ImageDesc {
  int obj_num = -1;
  int width = 0, height = 0, bpc = 8, ncomp = 1;
  int codec = 2;  // 0 = raw samples after generic filters, 1 = JPEG (DCTDecode last), 2 = unsupported
  bool indexed = false, invert = false, cmyk = false, mask = false;
  std::string palette;  // indexed: base components per entry
  int palette_comps = 3;
  // generic filter chain in front of the samples / JPEG data
  std::vector<std::string> filters;
  std::vector<std::array<int, 5>> parms;  // predictor, colors, bpc, columns, earlychange
  size_t off = 0, len = 0;                // raw stream bytes in the file buffer
  std::shared_ptr<std::string> owned;     // stream bytes when they come from an object stream (never for images) / cache
  double x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  std::string unsupported_reason;
};

struct PageCache {
  bool analyzed = false;
  std::string text_visible, text_all;
  std::string lines_visible, lines_all;   // positioned line records (assemble_lines)
  std::vector<ImageDesc> images;
  size_t chars_visible = 0, chars_invisible = 0, unmapped = 0, decoded = 0;
  double coverage = 0;
  std::string error;
};

struct Handle {
  std::unique_ptr<Document> doc;
  std::vector<PageCache> pages;
  std::mutex mu;
};

size_t utf8_len_nows(const std::string& s) {
  size_t n = 0;
  for (unsigned char c : s)
    if ((c & 0xC0) != 0x80 && c != ' ' && c != '\n' && c != '\t' && c != '\r') ++n;
  return n;
}

void describe_colorspace(Document& doc, ObjP cs, ImageDesc& im, int depth = 0) {
  cs = doc.resolve(cs);
  if (!cs || depth > 4) return;
  if (cs->type == Obj::Name) {
    const std::string& n = cs->s;
    if (n == "DeviceGray" || n == "G" || n == "CalGray") im.ncomp = 1;
    else if (n == "DeviceRGB" || n == "RGB" || n == "CalRGB") im.ncomp = 3;
    else if (n == "DeviceCMYK" || n == "CMYK") { im.ncomp = 4; im.cmyk = true; }
    return;
  }
  if (cs->type != Obj::Arr || cs->arr.empty()) return;
  ObjP head = doc.resolve(cs->arr[0]);
  if (!head || head->type != Obj::Name) return;
  const std::string& n = head->s;
  if (n == "ICCBased" && cs->arr.size() >= 2) {
    ObjP prof = doc.resolve(cs->arr[1]);
    int nc = (int)to_int(doc.dget(prof, "N"), 3);
    im.ncomp = nc;
    im.cmyk = nc == 4;
  } else if (n == "CalGray") im.ncomp = 1;
  else if (n == "CalRGB" || n == "Lab") im.ncomp = 3;
  else if ((n == "Indexed" || n == "I") && cs->arr.size() >= 4) {
    ImageDesc base;
    describe_colorspace(doc, cs->arr[1], base, depth + 1);
    im.indexed = true;
    im.palette_comps = base.ncomp;
    im.cmyk = false;
    ObjP lut = doc.resolve(cs->arr[3]);
    if (lut && lut->type == Obj::Str) im.palette = lut->s;
    else if (lut && lut->type == Obj::Stream) im.palette = doc.stream_data(lut);
    im.ncomp = 1;
  } else if (n == "Separation" || n == "DeviceN") {
    im.ncomp = n == "Separation" ? 1 : (int)(cs->arr.size() > 1 && doc.resolve(cs->arr[1]) ? doc.resolve(cs->arr[1])->arr.size() : 1);
    im.invert = n == "Separation";  // tint 1 = full ink
  }
}

ImageDesc describe_image(Document& doc, int obj_num, const ImagePlacement& pl) {
  ImageDesc im;
  im.obj_num = obj_num;
  im.x0 = pl.x0; im.y0 = pl.y0; im.x1 = pl.x1; im.y1 = pl.y1;
  ObjP x = obj_num >= 0 ? doc.load_object(obj_num) : nullptr;
  if (!x || x->type != Obj::Stream) { im.unsupported_reason = "inline or missing image"; return im; }
  im.width = (int)to_int(doc.dget(x, "Width"));
  im.height = (int)to_int(doc.dget(x, "Height"));
  im.bpc = (int)to_int(doc.dget(x, "BitsPerComponent"), 8);
  ObjP mk_ = doc.dget(x, "ImageMask");
  im.mask = mk_ && mk_->type == Obj::Bool && mk_->b;
  if (im.mask) { im.bpc = 1; im.ncomp = 1; }
  else describe_colorspace(doc, x->get("ColorSpace"), im);
  ObjP dec = doc.dget(x, "Decode");
  if (dec && dec->type == Obj::Arr && dec->arr.size() >= 2) {
    ObjP d0 = doc.resolve(dec->arr[0]), d1 = doc.resolve(dec->arr[1]);
    if (d0 && d1 && d0->is_num() && d1->is_num() && d0->num() > d1->num()) im.invert = !im.invert;
  }
  if (im.mask) im.invert = !im.invert;  // stencil: sample 0 paints (ink) under the default Decode
  if (x->owned) { im.owned = x->owned; }
  im.off = x->stream_off;
  im.len = x->stream_len;
  std::vector<std::string> filters;
  std::vector<ObjP> parms;
  ObjP f = doc.dget(x, "Filter");
  ObjP dp = doc.dget(x, "DecodeParms");
  if (!dp) dp = doc.dget(x, "DP");
  if (f) {
    if (f->type == Obj::Name) { filters.push_back(f->s); parms.push_back(dp); }
    else if (f->type == Obj::Arr)
      for (size_t i = 0; i < f->arr.size(); ++i) {
        ObjP fi = doc.resolve(f->arr[i]);
        filters.push_back(fi ? fi->s : "");
        parms.push_back((dp && dp->type == Obj::Arr && i < dp->arr.size()) ? doc.resolve(dp->arr[i]) : nullptr);
      }
  }
  im.codec = 0;
  for (size_t i = 0; i < filters.size(); ++i) {
    const std::string& fn = filters[i];
    if (fn == "DCTDecode" || fn == "DCT") {
      if (i + 1 != filters.size()) { im.codec = 2; im.unsupported_reason = "DCTDecode is not the last filter"; break; }
      im.codec = 1;
      break;
    }
    if (fn == "FlateDecode" || fn == "Fl" || fn == "LZWDecode" || fn == "LZW" || fn == "ASCIIHexDecode" || fn == "AHx" ||
        fn == "ASCII85Decode" || fn == "A85" || fn == "RunLengthDecode" || fn == "RL") {
      std::array<int, 5> pr = {1, 1, 8, 1, 1};
      if (parms[i] && parms[i]->type == Obj::Dict) {
        pr[0] = (int)to_int(doc.dget(parms[i], "Predictor"), 1);
        pr[1] = (int)to_int(doc.dget(parms[i], "Colors"), 1);
        pr[2] = (int)to_int(doc.dget(parms[i], "BitsPerComponent"), 8);
        pr[3] = (int)to_int(doc.dget(parms[i], "Columns"), 1);
        pr[4] = (int)to_int(doc.dget(parms[i], "EarlyChange"), 1);
      }
      im.filters.push_back(fn);
      im.parms.push_back(pr);
      continue;
    }
    if (fn == "CCITTFaxDecode" || fn == "CCF") {
      // K, Columns, Rows, BlackIs1, EncodedByteAlign (defaults of the PDF specification)
      std::array<int, 5> pr = {0, 1728, 0, 0, 0};
      if (parms[i] && parms[i]->type == Obj::Dict) {
        pr[0] = (int)to_int(doc.dget(parms[i], "K"), 0);
        pr[1] = (int)to_int(doc.dget(parms[i], "Columns"), 1728);
        pr[2] = (int)to_int(doc.dget(parms[i], "Rows"), 0);
        ObjP b1 = doc.dget(parms[i], "BlackIs1"), ba = doc.dget(parms[i], "EncodedByteAlign");
        pr[3] = b1 && b1->type == Obj::Bool && b1->b ? 1 : 0;
        pr[4] = ba && ba->type == Obj::Bool && ba->b ? 1 : 0;
      }
      im.filters.push_back("CCITTFaxDecode");
      im.parms.push_back(pr);
      continue;
    }
    im.codec = 2;
    im.unsupported_reason = "image codec " + fn + " is not supported";
    break;
  }
  if (im.codec == 0 && !(im.bpc == 1 || im.bpc == 2 || im.bpc == 4 || im.bpc == 8 || im.bpc == 16)) {
    im.codec = 2;
    im.unsupported_reason = "unsupported BitsPerComponent";
  }
  if (im.width <= 0 || im.height <= 0 || (long long)im.width * im.height > (1LL << 30)) {
    im.codec = 2;
    im.unsupported_reason = "bad image dimensions";
  }
  return im;
}

// Applies the generic filter chain; thread-safe (touches only the immutable file buffer).
std::string run_filters(const uint8_t* file, const ImageDesc& im) {
  const uint8_t* p = im.owned ? (const uint8_t*)im.owned->data() : file + im.off;
  size_t len = im.owned ? im.owned->size() : im.len;
  std::string cur;
  bool have = false;
  for (size_t i = 0; i < im.filters.size(); ++i) {
    const std::string& fn = im.filters[i];
    const uint8_t* ip = have ? (const uint8_t*)cur.data() : p;
    size_t il = have ? cur.size() : len;
    std::string next;
    if (fn == "CCITTFaxDecode") {
      const auto& pr = im.parms[i];
      const int rows = pr[2] > 0 ? pr[2] : im.height;
      next = ccitt::decode(ip, il, pr[0], pr[1], rows, pr[3] != 0, pr[4] != 0, im.height > 0 ? im.height : (1 << 16));
      cur.swap(next);
      have = true;
      continue;
    }
    if (fn == "FlateDecode" || fn == "Fl") {
      size_t hint = ((size_t)im.width * im.ncomp * im.bpc + 7) / 8 * im.height + im.height + 64;
      next = inflate_bytes(ip, il, i + 1 == im.filters.size() ? hint : 0);
    } else if (fn == "LZWDecode" || fn == "LZW") next = lzw_decode(ip, il, im.parms[i][4]);
    else if (fn == "ASCIIHexDecode" || fn == "AHx") next = ascii_hex_decode(ip, il);
    else if (fn == "ASCII85Decode" || fn == "A85") next = ascii85_decode(ip, il);
    else next = runlength_decode(ip, il);
    if (im.parms[i][0] > 1) next = apply_predictor(std::move(next), im.parms[i][0], im.parms[i][1], im.parms[i][2], im.parms[i][3]);
    cur.swap(next);
    have = true;
  }
  if (!have) cur.assign((const char*)p, len);
  return cur;
}

void analyze_page(Handle* h, int page) {
  PageCache& pc = h->pages[page];
  if (pc.analyzed) return;
  pc.analyzed = true;
  try {
    PageContent content;
    ContentInterpreter ci(*h->doc, content);
    const auto& pg = h->doc->pages()[page];
    ci.run_page(pg);
    pc.text_visible = assemble_text(content.spans, false);
    pc.text_all = assemble_text(content.spans, true);
    pc.lines_visible = assemble_lines(content.spans, false);
    pc.lines_all = assemble_lines(content.spans, true);
    pc.chars_visible = utf8_len_nows(pc.text_visible);
    pc.chars_invisible = utf8_len_nows(pc.text_all) - pc.chars_visible;
    pc.unmapped = content.unmapped;
    pc.decoded = content.decoded;
    double pw = pg.mediabox[2] - pg.mediabox[0], ph = pg.mediabox[3] - pg.mediabox[1];
    double area = std::max(1.0, pw * ph), cov = 0;
    for (const auto& ip : content.images) {
      double x0 = std::max(ip.x0, pg.mediabox[0]), x1 = std::min(ip.x1, pg.mediabox[2]);
      double y0 = std::max(ip.y0, pg.mediabox[1]), y1 = std::min(ip.y1, pg.mediabox[3]);
      if (x1 > x0 && y1 > y0) cov += (x1 - x0) * (y1 - y0);
      pc.images.push_back(describe_image(*h->doc, ip.obj_num, ip));
    }
    pc.coverage = std::min(1.0, cov / area);
  } catch (const std::exception& e) {
    pc.error = e.what();
  }
}

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

}  // namespace

extern "C" {

struct DlpdfPageInfo {
  double width_pt, height_pt;
  int32_t rotate, n_images;
  int64_t chars_visible, chars_invisible, unmapped, decoded;
  double image_coverage;
  int32_t has_error, pad;
};

struct DlpdfImageInfo {
  int32_t obj_num, width, height, bpc, ncomp, codec, out_channels, pad;
  int64_t stream_off, stream_len;
  double x0, y0, x1, y1;
};

static void set_err(char* err, int errlen, const char* msg) {
  if (err && errlen > 0) { strncpy(err, msg, errlen - 1); err[errlen - 1] = 0; }
}

// The caller keeps ``data`` alive and unmodified until dlpdf_close.
void* dlpdf_open(const uint8_t* data, size_t n, char* err, int errlen) {
  try {
    auto h = std::make_unique<Handle>();
    h->doc = std::make_unique<Document>(data, n);
    h->pages.resize(h->doc->pages().size());
    return h.release();
  } catch (const std::exception& e) {
    set_err(err, errlen, e.what());
    return nullptr;
  } catch (...) {
    set_err(err, errlen, "unknown error while opening PDF");
    return nullptr;
  }
}

void dlpdf_close(void* hp) { delete (Handle*)hp; }

int dlpdf_page_count(void* hp) { return hp ? (int)((Handle*)hp)->pages.size() : -1; }

int dlpdf_was_rebuilt(void* hp) { return hp && ((Handle*)hp)->doc->was_rebuilt() ? 1 : 0; }

int dlpdf_page_info(void* hp, int page, DlpdfPageInfo* out) {
  Handle* h = (Handle*)hp;
  if (!h || page < 0 || (size_t)page >= h->pages.size()) return -1;
  std::lock_guard<std::mutex> g(h->mu);
  analyze_page(h, page);
  const PageCache& pc = h->pages[page];
  const auto& pg = h->doc->pages()[page];
  out->width_pt = pg.mediabox[2] - pg.mediabox[0];
  out->height_pt = pg.mediabox[3] - pg.mediabox[1];
  out->rotate = pg.rotate;
  out->n_images = (int)pc.images.size();
  out->chars_visible = (int64_t)pc.chars_visible;
  out->chars_invisible = (int64_t)pc.chars_invisible;
  out->unmapped = (int64_t)pc.unmapped;
  out->decoded = (int64_t)pc.decoded;
  out->image_coverage = pc.coverage;
  out->has_error = pc.error.empty() ? 0 : 1;
  out->pad = 0;
  return 0;
}

// Pointer stays valid until dlpdf_close.
int dlpdf_page_text(void* hp, int page, int include_invisible, const char** out, size_t* len) {
  Handle* h = (Handle*)hp;
  if (!h || page < 0 || (size_t)page >= h->pages.size()) return -1;
  std::lock_guard<std::mutex> g(h->mu);
  analyze_page(h, page);
  const std::string& t = include_invisible ? h->pages[page].text_all : h->pages[page].text_visible;
  *out = t.data();
  *len = t.size();
  return 0;
}

// Positioned text lines of a page: "x0\ty0\tx1\ty1\tsize\ttext\n" records in page space (y up).  Pointer stays valid
// until dlpdf_close.
int dlpdf_page_lines(void* hp, int page, int include_invisible, const char** out, size_t* len) {
  Handle* h = (Handle*)hp;
  if (!h || page < 0 || (size_t)page >= h->pages.size()) return -1;
  std::lock_guard<std::mutex> g(h->mu);
  analyze_page(h, page);
  const std::string& t = include_invisible ? h->pages[page].lines_all : h->pages[page].lines_visible;
  *out = t.data();
  *len = t.size();
  return 0;
}

int dlpdf_page_error(void* hp, int page, const char** out, size_t* len) {
  Handle* h = (Handle*)hp;
  if (!h || page < 0 || (size_t)page >= h->pages.size()) return -1;
  std::lock_guard<std::mutex> g(h->mu);
  *out = h->pages[page].error.data();
  *len = h->pages[page].error.size();
  return 0;
}

int dlpdf_image_info(void* hp, int page, int idx, DlpdfImageInfo* out) {
  Handle* h = (Handle*)hp;
  if (!h || page < 0 || (size_t)page >= h->pages.size()) return -1;
  std::lock_guard<std::mutex> g(h->mu);
  analyze_page(h, page);
  const PageCache& pc = h->pages[page];
  if (idx < 0 || (size_t)idx >= pc.images.size()) return -2;
  const ImageDesc& im = pc.images[idx];
  out->obj_num = im.obj_num;
  out->width = im.width; out->height = im.height; out->bpc = im.bpc; out->ncomp = im.ncomp;
  out->codec = im.codec;
  bool color = (!im.indexed && (im.ncomp == 3 || im.ncomp == 4)) || (im.indexed && im.palette_comps >= 3);
  out->out_channels = color ? 3 : 1;
  out->pad = 0;
  // a JPEG that is the only filter can be handed to the decoder straight from the file buffer
  bool direct = im.codec == 1 && im.filters.empty() && !im.owned;
  out->stream_off = direct ? (int64_t)im.off : -1;
  out->stream_len = direct ? (int64_t)im.len : -1;
  out->x0 = im.x0; out->y0 = im.y0; out->x1 = im.x1; out->y1 = im.y1;
  return 0;
}

// Encoded JPEG bytes of an image whose DCT data sits behind other filters (rare).  Caller frees with dlpdf_free.
int dlpdf_image_jpeg_bytes(void* hp, int page, int idx, uint8_t** out, size_t* len) {
  Handle* h = (Handle*)hp;
  if (!h || page < 0 || (size_t)page >= h->pages.size()) return -1;
  ImageDesc im;
  {
    std::lock_guard<std::mutex> g(h->mu);
    analyze_page(h, page);
    const PageCache& pc = h->pages[page];
    if (idx < 0 || (size_t)idx >= pc.images.size()) return -2;
    im = pc.images[idx];
  }
  if (im.codec != 1) return -3;
  try {
    std::string data = run_filters(h->doc->data(), im);
    *out = (uint8_t*)malloc(data.size() ? data.size() : 1);
    if (!*out) return -4;
    memcpy(*out, data.data(), data.size());
    *len = data.size();
    return 0;
  } catch (const std::exception&) {
    return -5;
  }
}

void dlpdf_free(void* p) { free(p); }

// The in-repo zlib-stream decoder on its own (tests, benchmarks): bytes written to dst, or -1 when it refuses the
// stream (callers inside the library then use zlib).
long long dlpdf_fast_inflate(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
  return fastinflate::inflate_zlib(src, n, dst, cap);
}

// Decode a non-JPEG image to 8-bit gray (out_channels 1) or RGB (3), row-major, into dst.
// Thread-safe after the page has been analysed.  Returns bytes written or < 0.
long long dlpdf_image_decode(void* hp, int page, int idx, uint8_t* dst, size_t dst_len) {
  Handle* h = (Handle*)hp;
  if (!h || page < 0 || (size_t)page >= h->pages.size()) return -1;
  ImageDesc im;
  {
    std::lock_guard<std::mutex> g(h->mu);
    analyze_page(h, page);
    const PageCache& pc = h->pages[page];
    if (idx < 0 || (size_t)idx >= pc.images.size()) return -2;
    im = pc.images[idx];
  }
  if (im.codec != 0) return -3;
  try {
    // the common scanner layout - one FlateDecode filter, no predictor, 8-bit gray or RGB - is inflated straight into
    // the caller's (pinned) buffer: no intermediate string, no second pass over the 9-26 MB of raster
    if (im.filters.size() == 1 && (im.filters[0] == "FlateDecode" || im.filters[0] == "Fl") && im.parms[0][0] <= 1 &&
        im.bpc == 8 && !im.indexed && !im.invert && !im.mask && (im.ncomp == 1 || im.ncomp == 3)) {
      const size_t need = (size_t)im.width * im.height * im.ncomp;
      if (dst_len >= need) {
        const uint8_t* p = im.owned ? (const uint8_t*)im.owned->data() : h->doc->data() + im.off;
        const size_t len = im.owned ? im.owned->size() : im.len;
        const long long got = fastinflate::inflate_zlib(p, len, dst, need);
        if (got == (long long)need) return (long long)need;
        // short, long or damaged streams: the general path below pads / truncates / tolerates as before
      }
    }
    // the same layout written by image tools: rows PNG-predicted before deflating.  Inflate into a scratch buffer, undo
    // the row filters straight into the caller's buffer.
    if (im.filters.size() == 1 && (im.filters[0] == "FlateDecode" || im.filters[0] == "Fl") && im.parms[0][0] >= 10 &&
        im.bpc == 8 && !im.indexed && !im.invert && !im.mask && (im.ncomp == 1 || im.ncomp == 3) &&
        im.parms[0][1] == im.ncomp && im.parms[0][2] == 8 && im.parms[0][3] == im.width) {
      const size_t rowbytes = (size_t)im.width * im.ncomp, need = rowbytes * im.height;
      if (dst_len >= need) {
        const uint8_t* p = im.owned ? (const uint8_t*)im.owned->data() : h->doc->data() + im.off;
        const size_t len = im.owned ? im.owned->size() : im.len;
        static thread_local std::vector<uint8_t> scratch;
        const size_t packed = (rowbytes + 1) * im.height;
        if (scratch.size() < packed) scratch.resize(packed);
        if (fastinflate::inflate_zlib(p, len, scratch.data(), packed) == (long long)packed) {
          png_unpredict_rows(scratch.data(), im.height, rowbytes, im.ncomp, dst);
          return (long long)need;
        }
      }
    }
    std::string raw = run_filters(h->doc->data(), im);
    const int W = im.width, H = im.height, nc = im.ncomp, bpc = im.bpc;
    const bool color = (!im.indexed && (nc == 3 || nc == 4)) || (im.indexed && im.palette_comps >= 3);
    const int oc = color ? 3 : 1;
    const size_t need = (size_t)W * H * oc;
    if (dst_len < need) return -4;
    const size_t rowbytes = ((size_t)W * nc * bpc + 7) / 8;
    if (raw.size() < rowbytes * H) raw.resize(rowbytes * H, im.invert ? (char)0xFF : 0);  // truncated: pad with paper
    // fast path: 8-bit gray / RGB, no palette
    if (bpc == 8 && !im.indexed && !im.invert && (nc == 1 || nc == 3)) {
      memcpy(dst, raw.data(), need);
      return (long long)need;
    }
    // bilevel pages (CCITT fax, 1-bit Flate): a table turns every input byte into its eight output pixels at once
    // (the general loop below spends ~6 ns per pixel, 50 ms for an A4 page at 300 DPI)
    if (bpc == 1 && nc == 1 && !im.indexed) {
      uint64_t lut[256];
      const uint8_t on = im.invert ? 0 : 255, off = im.invert ? 255 : 0;
      for (int v = 0; v < 256; ++v) {
        uint8_t px[8];
        for (int k = 0; k < 8; ++k) px[k] = (v >> (7 - k)) & 1 ? on : off;
        memcpy(&lut[v], px, 8);
      }
      const int whole = W >> 3;
      for (int y = 0; y < H; ++y) {
        const uint8_t* row = (const uint8_t*)raw.data() + (size_t)y * rowbytes;
        uint8_t* o = dst + (size_t)y * W;
        for (int b = 0; b < whole; ++b) memcpy(o + 8 * b, &lut[row[b]], 8);
        for (int x = whole << 3; x < W; ++x) o[x] = (row[x >> 3] >> (7 - (x & 7))) & 1 ? on : off;
      }
      return (long long)need;
    }
    const int maxv = (1 << std::min(bpc, 8)) - 1;
    std::vector<int> comp(std::max(nc, 4));
    for (int y = 0; y < H; ++y) {
      const uint8_t* row = (const uint8_t*)raw.data() + (size_t)y * rowbytes;
      uint8_t* o = dst + (size_t)y * W * oc;
      size_t bit = 0;
      for (int x = 0; x < W; ++x) {
        for (int c = 0; c < nc; ++c) {
          int v;
          if (bpc == 8) v = row[(size_t)x * nc + c];
          else if (bpc == 16) v = row[((size_t)x * nc + c) * 2];
          else {
            v = (row[bit >> 3] >> (8 - bpc - (bit & 7))) & maxv;
            bit += bpc;
          }
          comp[c] = v;
        }
        if (im.indexed) {
          int pcn = im.palette_comps;
          size_t base = (size_t)comp[0] * pcn;
          int pv[4] = {0, 0, 0, 0};
          for (int c = 0; c < pcn && c < 4; ++c) pv[c] = base + c < im.palette.size() ? (uint8_t)im.palette[base + c] : 0;
          if (pcn == 1) o[x] = (uint8_t)pv[0];
          else if (pcn == 3) { o[3 * x] = pv[0]; o[3 * x + 1] = pv[1]; o[3 * x + 2] = pv[2]; }
          else { for (int c = 0; c < 3; ++c) o[3 * x + c] = clamp8(255 - pv[c] - pv[3]); }
          continue;
        }
        if (bpc < 8) for (int c = 0; c < nc; ++c) comp[c] = comp[c] * 255 / maxv;
        if (im.invert) for (int c = 0; c < nc; ++c) comp[c] = 255 - comp[c];
        if (nc == 1) o[x] = (uint8_t)comp[0];
        else if (nc == 3) { o[3 * x] = comp[0]; o[3 * x + 1] = comp[1]; o[3 * x + 2] = comp[2]; }
        else if (nc == 4) { for (int c = 0; c < 3; ++c) o[3 * x + c] = clamp8(255 - comp[c] - comp[3]); }
        else o[x] = (uint8_t)comp[0];
      }
    }
    return (long long)need;
  } catch (const std::exception&) {
    return -5;
  }
}

// Bilevel images (CCITT fax, 1-bit Flate) as PACKED bits: rows of (width + 7) / 8 bytes, most significant bit first, a
// set bit = white (255) once /Decode inversion is applied.  The caller uploads the 1.1 MB of an A4 page at 300 DPI
// and expands it on the GPU instead of writing and uploading 8.7 MB of bytes.  Returns the byte count, or a negative
// code (-6: not a plain bilevel image - use dlpdf_image_decode).
long long dlpdf_image_decode_bits(void* hp, int page, int idx, uint8_t* dst, size_t dst_len) {
  Handle* h = (Handle*)hp;
  if (!h || page < 0 || (size_t)page >= h->pages.size()) return -1;
  ImageDesc im;
  {
    std::lock_guard<std::mutex> g(h->mu);
    analyze_page(h, page);
    const PageCache& pc = h->pages[page];
    if (idx < 0 || (size_t)idx >= pc.images.size()) return -2;
    im = pc.images[idx];
  }
  if (im.codec != 0) return -3;
  if (im.bpc != 1 || im.ncomp != 1 || im.indexed || im.mask) return -6;
  try {
    std::string raw = run_filters(h->doc->data(), im);
    const size_t rowbytes = ((size_t)im.width + 7) / 8, need = rowbytes * im.height;
    if (dst_len < need) return -4;
    if (raw.size() < need) raw.resize(need, im.invert ? (char)0xFF : 0);     // truncated: pad with paper
    if (im.invert) {
      for (size_t k = 0; k < need; ++k) dst[k] = (uint8_t)~(uint8_t)raw[k];
    } else {
      memcpy(dst, raw.data(), need);
    }
    return (long long)need;
  } catch (const std::exception&) {
    return -5;
  }
}

struct DlpdfClassify {
  int32_t n_pages, is_text;          // is_text: 1 = born-digital (text branch), 0 = needs OCR
  int32_t pages_with_text, pages_image_dominant;
  double avg_chars, garbled_ratio, avg_coverage;
};

// Document-level TXT-vs-OCR decision (the role of magic-pdf's ``classify()``, ocr_service.py:213):
//   text  <=> enough visible characters per page, few undecodable codes, and the pages are not
//             dominated by full-page images (a scan with a hidden text layer still goes to OCR).
int dlpdf_classify(void* hp, DlpdfClassify* out) {
  Handle* h = (Handle*)hp;
  if (!h) return -1;
  std::lock_guard<std::mutex> g(h->mu);
  const int n = (int)h->pages.size();
  long long chars = 0, unmapped = 0, decoded = 0;
  int with_text = 0, img_dom = 0;
  double cov = 0;
  for (int i = 0; i < n; ++i) {
    analyze_page(h, i);
    const PageCache& pc = h->pages[i];
    chars += (long long)pc.chars_visible;
    unmapped += (long long)pc.unmapped;
    decoded += (long long)pc.decoded;
    cov += pc.coverage;
    if (pc.chars_visible >= 20) ++with_text;
    if (pc.coverage >= 0.5 && pc.chars_visible < 300) ++img_dom;
  }
  out->n_pages = n;
  out->pages_with_text = with_text;
  out->pages_image_dominant = img_dom;
  out->avg_chars = n ? (double)chars / n : 0.0;
  out->garbled_ratio = (unmapped + decoded) ? (double)unmapped / (double)(unmapped + decoded) : 0.0;
  out->avg_coverage = n ? cov / n : 0.0;
  // short born-digital documents (a one-line letter) have few characters but no page image at all
  bool enough = out->avg_chars >= 100.0 || (img_dom == 0 && chars > 0 && out->avg_coverage < 0.2);
  bool text = n > 0 && enough && out->garbled_ratio < 0.05 && img_dom * 2 < std::max(n, 1);
  out->is_text = text ? 1 : 0;
  return 0;
}

// ------------------------------------------------------------------ JPEG header scan (no PDF handle needed)
int dljpeg_sizes(int* info_size, int* tables_size) {
  *info_size = (int)sizeof(dljpeg::JpegInfo);
  *tables_size = (int)sizeof(dljpeg::JpegTables);
  return 0;
}

// Host emulation of the GPU's self-synchronising decode (tests).  out: gray plane [mcus_y*vs0*8][mcus_x*hs0*8].
int dljpeg_decode_sync_host(const uint8_t* d, size_t n, uint8_t* out, size_t cap, dljpeg::SyncStats* stats) {
  try {
    return dljpeg::decode_sync_host(d, n, out, cap, stats);
  } catch (...) {
    return -100;
  }
}

// seg_offsets may be null (count only).  Returns 0, or < 0 on malformed input; info->n_segments holds the
// number of restart segments found (may exceed max_segs, in which case only max_segs were written).
int dljpeg_scan(const uint8_t* d, size_t n, dljpeg::JpegInfo* info, dljpeg::JpegTables* tables, uint32_t* seg_offsets,
                int max_segs) {
  try {
    std::vector<uint32_t> segs;
    int rc = dljpeg::scan_jpeg(d, n, info, tables, &segs);
    if (rc != 0) return rc;
    if (seg_offsets) {
      size_t k = std::min<size_t>(segs.size(), (size_t)std::max(0, max_segs));
      memcpy(seg_offsets, segs.data(), k * sizeof(uint32_t));
    }
    return 0;
  } catch (...) {
    return -100;
  }
}

}  // extern "C"
