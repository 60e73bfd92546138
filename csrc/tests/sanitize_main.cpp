// Host-code sanitizer harness: the native PDF reader, JPEG scanner, dynamic batcher and BPE
// vocabulary compiled with -fsanitize=address,undefined and driven with (a) well-formed documents given on the
// command line and (b) thousands of mutated / truncated copies of them.  Any heap overflow, use-after-free,
// signed overflow or misaligned access aborts the run; malformed input must only ever produce error codes.
//
//   python -m distributed_llms_amd.utils.sanitize            (builds build/sanitize/dlamd_san and runs it)
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <zlib.h>

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
struct DlpdfClassify {
  int32_t n_pages, is_text;
  int32_t pages_with_text, pages_image_dominant;
  double avg_chars, garbled_ratio, avg_coverage;
};
void* dlpdf_open(const uint8_t* data, size_t n, char* err, int errlen);
void dlpdf_close(void* h);
int dlpdf_page_count(void* h);
int dlpdf_page_info(void* h, int page, DlpdfPageInfo* out);
int dlpdf_page_text(void* h, int page, int include_invisible, const char** out, size_t* len);
int dlpdf_image_info(void* h, int page, int idx, DlpdfImageInfo* out);
int dlpdf_image_jpeg_bytes(void* h, int page, int idx, uint8_t** out, size_t* len);
void dlpdf_free(void* p);
long long dlpdf_image_decode(void* h, int page, int idx, uint8_t* dst, size_t dst_len);
int dlpdf_classify(void* h, DlpdfClassify* out);
long long dlpdf_fast_inflate(const uint8_t* src, size_t n, uint8_t* dst, size_t cap);
int dljpeg_sizes(int* info_size, int* tables_size);
int dljpeg_scan(const uint8_t* d, size_t n, void* info, void* tables, uint32_t* seg_offsets, int max_segs);

void* dlrt_batcher_create(uint64_t budget, int max_items, int64_t max_wait_us);
void dlrt_batcher_destroy(void* h);
int dlrt_batcher_push(void* h, uint64_t key, uint64_t id, uint64_t cost);
void dlrt_batcher_close(void* h);
int dlrt_batcher_pop(void* h, int flush, int64_t timeout_us, uint64_t* ids, int cap, uint64_t* key_out);
int dlrt_batcher_pending(void* h);

void* dlbpe_create_bytes();
void* dlbpe_create(const char* vocab_json, long long n, const char* merges_txt, long long m);
void dlbpe_destroy(void* h);
long long dlbpe_decode_batch(void* h, const int32_t* ids, int rows, int cols, int stop_at_eos, char* out, long long cap,
                             long long* offsets);
long long dlbpe_encode(void* h, const char* utf8, long long n, int32_t* out_ids, long long cap);
}

static std::vector<uint8_t> read_file(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) return v;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize(n > 0 ? (size_t)n : 0);
  if (n > 0 && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
  fclose(f);
  return v;
}

// Walks everything the Python binding can reach. Returns pages seen (0 when the open was refused).
static int walk(const std::vector<uint8_t>& pdf, long long* chars, int* images) {
  char err[256] = {0};
  void* h = dlpdf_open(pdf.data(), pdf.size(), err, sizeof err);
  if (!h) return 0;
  const int n = dlpdf_page_count(h);
  int isz = 0, tsz = 0;
  dljpeg_sizes(&isz, &tsz);
  std::vector<uint8_t> jinfo(isz), jtab(tsz);
  for (int p = 0; p < n && p < 64; ++p) {
    DlpdfPageInfo pi;
    if (dlpdf_page_info(h, p, &pi) != 0) continue;
    const char* t = nullptr;
    size_t tl = 0;
    if (dlpdf_page_text(h, p, 1, &t, &tl) == 0) {
      unsigned acc = 0;
      for (size_t i = 0; i < tl; ++i) acc += (unsigned char)t[i];   // touch every byte
      *chars += (long long)tl + (acc & 0);
    }
    for (int k = 0; k < pi.n_images && k < 16; ++k) {
      DlpdfImageInfo ii;
      if (dlpdf_image_info(h, p, k, &ii) != 0) continue;
      ++*images;
      if (ii.codec == 1) {
        uint8_t* jb = nullptr;
        size_t jl = 0;
        if (dlpdf_image_jpeg_bytes(h, p, k, &jb, &jl) == 0 && jb) {
          std::vector<uint32_t> segs(4096);
          dljpeg_scan(jb, jl, jinfo.data(), jtab.data(), segs.data(), (int)segs.size());
          dlpdf_free(jb);
        }
      } else if (ii.codec == 0 && ii.width > 0 && ii.height > 0 && (long long)ii.width * ii.height < (1LL << 26)) {
        std::vector<uint8_t> dst((size_t)ii.width * ii.height * (ii.out_channels > 0 ? ii.out_channels : 1));
        dlpdf_image_decode(h, p, k, dst.data(), dst.size());
        if (dst.size() > 8) dlpdf_image_decode(h, p, k, dst.data(), dst.size() / 2);   // short buffer must be refused
      }
    }
  }
  DlpdfClassify c;
  dlpdf_classify(h, &c);
  dlpdf_close(h);
  return n;
}

static void fuzz_pdf(const std::vector<uint8_t>& seed, int iters, uint32_t rng_seed) {
  std::mt19937 rng(rng_seed);
  long long chars = 0;
  int images = 0, opened = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> m = seed;
    const int mode = (int)(rng() % 5);
    if (mode == 0 && m.size() > 16) m.resize(rng() % m.size());                       // truncate
    else if (mode == 1) for (int k = 0; k < 1 + (int)(rng() % 8); ++k) m[rng() % m.size()] = (uint8_t)rng();   // byte flips
    else if (mode == 2 && m.size() > 64) {                                              // splice a window
      size_t a = rng() % m.size(), b = rng() % m.size(), l = rng() % 64;
      for (size_t k = 0; k < l && a + k < m.size() && b + k < m.size(); ++k) m[a + k] = m[b + k];
    } else if (mode == 3) {                                                             // digit corruption (lengths, offsets, refs)
      for (int k = 0; k < 6; ++k) { size_t p = rng() % m.size(); if (m[p] >= '0' && m[p] <= '9') m[p] = '0' + rng() % 10; }
    } else if (m.size() > 32) {                                                         // drop a slice
      size_t a = rng() % m.size(), l = 1 + rng() % 32;
      m.erase(m.begin() + a, m.begin() + std::min(m.size(), a + l));
    }
    opened += walk(m, &chars, &images) > 0;
  }
  printf("  fuzz: %d variants, %d still opened, %lld text bytes, %d images walked\n", iters, opened, chars, images);
}

static void batcher_stress() {
  void* b = dlrt_batcher_create(1 << 20, 8, 2000);
  std::atomic<long long> popped{0};
  std::atomic<bool> closed{false};
  std::vector<std::thread> th;
  for (int p = 0; p < 4; ++p)
    th.emplace_back([b, p] { for (int i = 0; i < 5000; ++i) dlrt_batcher_push(b, (uint64_t)(i % 5), (uint64_t)(p * 100000 + i + 1), 1000 + i % 7); });
  for (int c = 0; c < 2; ++c)
    th.emplace_back([b, &popped, &closed] {
      uint64_t ids[8], key;
      for (;;) {
        const bool was_closed = closed.load();
        const int n = dlrt_batcher_pop(b, 0, 5000, ids, 8, &key);   // 0 = timeout, or closed and drained
        popped += n;
        if (n == 0 && was_closed) break;
      }
    });
  for (int p = 0; p < 4; ++p) th[p].join();
  while (dlrt_batcher_pending(b) > 0) std::this_thread::sleep_for(std::chrono::milliseconds(2));
  dlrt_batcher_close(b);
  closed.store(true);
  for (size_t i = 4; i < th.size(); ++i) th[i].join();
  dlrt_batcher_destroy(b);
  printf("  batcher: 20000 pushed, %lld popped\n", popped.load());
  if (popped.load() != 20000) { fprintf(stderr, "batcher lost items\n"); exit(3); }
}

static void bpe_stress() {
  void* v = dlbpe_create_bytes();
  std::mt19937 rng(7);
  for (int it = 0; it < 2000; ++it) {
    std::string s(rng() % 200, 0);
    for (auto& c : s) c = (char)rng();
    std::vector<int32_t> ids(s.size() + 4);
    long long n = dlbpe_encode(v, s.data(), (long long)s.size(), ids.data(), (long long)ids.size());
    std::vector<char> out(s.size() + 8);
    long long off[2];
    long long m = dlbpe_decode_batch(v, ids.data(), 1, (int)n, 1, out.data(), (long long)out.size(), off);
    if (n != (long long)s.size() || m != n || memcmp(out.data(), s.data(), s.size()) != 0) { fprintf(stderr, "bpe roundtrip\n"); exit(4); }
    for (auto& i : ids) i = (int32_t)rng();                               // wild ids, tiny output buffer
    dlbpe_decode_batch(v, ids.data(), 1, (int)ids.size(), 0, out.data(), 3, off);
  }
  dlbpe_destroy(v);
  const std::string good = "{\"<s>\":0,\"</s>\":1,\"a\":2,\"\\u0120\":3,\"\\u0120a\":4,\"\\ud83d\\ude00\":5}";
  for (int it = 0; it < 4000; ++it) {
    std::string j = good;
    for (int k = 0; k < (int)(rng() % 4); ++k) j[rng() % j.size()] = (char)rng();
    if (rng() % 3 == 0) j.resize(rng() % j.size());
    void* h = dlbpe_create(j.data(), (long long)j.size(), "\xc4\xa0 a\n", 5);
    if (h) {
      int32_t ids[16];
      dlbpe_encode(h, "a a  a", 6, ids, 16);
      dlbpe_destroy(h);
    }
  }
  printf("  bpe: round trips and malformed vocabularies ok\n");
}

// Hand-built JPEG headers aimed at the table builders: an over-subscribed DHT (more codewords of a length than
// a prefix code can hold), Huffman selectors outside huff[2][4] in SOS, a scan component the frame never declared,
// and random mutations of the DHT / SOS payload bytes of an otherwise valid header.
static void put_seg(std::vector<uint8_t>& j, uint8_t marker, const std::vector<uint8_t>& body) {
  j.push_back(0xFF); j.push_back(marker);
  const size_t len = body.size() + 2;
  j.push_back((uint8_t)(len >> 8)); j.push_back((uint8_t)len);
  j.insert(j.end(), body.begin(), body.end());
}

static std::vector<uint8_t> craft_jpeg(const uint8_t* dht_bits, int n_vals, uint8_t sos_sel, uint8_t sos_cid) {
  std::vector<uint8_t> j = {0xFF, 0xD8};
  std::vector<uint8_t> dqt(65, 1); dqt[0] = 0;
  put_seg(j, 0xDB, dqt);
  put_seg(j, 0xC0, {8, 0, 16, 0, 16, 1, 1, 0x11, 0});
  for (int tc = 0; tc < 2; ++tc) {
    std::vector<uint8_t> dht; dht.push_back((uint8_t)(tc << 4));
    for (int k = 1; k <= 16; ++k) dht.push_back(dht_bits[k]);
    for (int k = 0; k < n_vals; ++k) dht.push_back((uint8_t)k);
    put_seg(j, 0xC4, dht);
  }
  put_seg(j, 0xDA, {1, sos_cid, sos_sel, 0, 63, 0});
  for (int k = 0; k < 64; ++k) j.push_back((uint8_t)(k * 37));
  j.push_back(0xFF); j.push_back(0xD9);
  return j;
}

static void jpeg_crafted() {
  int isz = 0, tsz = 0;
  dljpeg_sizes(&isz, &tsz);
  std::vector<uint8_t> jinfo(isz), jtab(tsz);
  std::vector<uint32_t> segs(64);
  auto scan = [&](const std::vector<uint8_t>& j) { return dljpeg_scan(j.data(), j.size(), jinfo.data(), jtab.data(), segs.data(), (int)segs.size()); };
  uint8_t ok_bits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};       // the standard DC-luma table
  if (scan(craft_jpeg(ok_bits, 12, 0x00, 1)) != 0) { fprintf(stderr, "crafted valid jpeg refused\n"); exit(5); }
  uint8_t over[17] = {0, 255, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};         // 255 one-bit codewords
  if (scan(craft_jpeg(over, 255, 0x00, 1)) == 0) { fprintf(stderr, "over-subscribed DHT accepted\n"); exit(5); }
  uint8_t over2[17] = {0, 2, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // full at length 1, one more at 2
  if (scan(craft_jpeg(over2, 3, 0x00, 1)) == 0) { fprintf(stderr, "over-subscribed DHT (len 2) accepted\n"); exit(5); }
  if (scan(craft_jpeg(ok_bits, 12, 0xFF, 1)) == 0) { fprintf(stderr, "SOS selector 0xFF accepted\n"); exit(5); }
  if (scan(craft_jpeg(ok_bits, 12, 0x00, 9)) == 0) { fprintf(stderr, "undeclared scan component accepted\n"); exit(5); }
  std::mt19937 rng(99);
  const std::vector<uint8_t> good = craft_jpeg(ok_bits, 12, 0x00, 1);
  int accepted = 0;
  for (int it = 0; it < 20000; ++it) {
    std::vector<uint8_t> m = good;
    for (int k = 0; k < 1 + (int)(rng() % 6); ++k) m[2 + rng() % (m.size() - 2)] = (uint8_t)rng();
    accepted += scan(m) == 0;
  }
  printf("  jpeg: crafted DHT/SOS headers refused, %d of 20000 mutated headers still parsed\n", accepted);
}

// The in-repo inflater against zlib's deflate: exact round trips at several compression settings, then truncated and
// bit-flipped streams with tight output buffers - it may refuse them (-1), it must never write past ``cap`` (the
// buffer is heap-allocated at exactly that size, so ASan sees one byte too many).
static void inflate_stress() {
  std::mt19937 rng(4242);
  std::vector<std::vector<uint8_t>> raws;
  raws.push_back(std::vector<uint8_t>(200000, 255));                           // paper
  std::vector<uint8_t> noisy(300000);
  for (size_t i = 0; i < noisy.size(); ++i) noisy[i] = (uint8_t)(((i / 700) % 5 == 0 ? 30 : 235) + (rng() % 9));
  raws.push_back(noisy);
  std::vector<uint8_t> rnd(70000);
  for (auto& b : rnd) b = (uint8_t)rng();
  raws.push_back(rnd);
  raws.push_back(std::vector<uint8_t>());
  long long refused = 0, total = 0;
  for (const auto& raw : raws)
    for (int level : {0, 1, 6, 9})
      for (int strategy : {Z_DEFAULT_STRATEGY, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED}) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, 15, 8, strategy) != Z_OK) exit(6);
        std::vector<uint8_t> z(deflateBound(&zs, (uLong)raw.size()) + 16);
        zs.next_in = const_cast<Bytef*>(raw.data());
        zs.avail_in = (uInt)raw.size();
        zs.next_out = z.data();
        zs.avail_out = (uInt)z.size();
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) exit(6);
        z.resize(zs.total_out);
        deflateEnd(&zs);
        {
          std::vector<uint8_t> out(raw.size() ? raw.size() : 1);
          const long long got = dlpdf_fast_inflate(z.data(), z.size(), out.data(), raw.size());
          if (got != (long long)raw.size() || (raw.size() && memcmp(out.data(), raw.data(), raw.size()))) {
            fprintf(stderr, "inflate round trip failed (level %d strategy %d size %zu got %lld)\n", level, strategy, raw.size(), got);
            exit(6);
          }
        }
        for (int it = 0; it < 60; ++it) {
          std::vector<uint8_t> m = z;
          const int kind = it % 3;
          if (kind == 0 && m.size() > 3) m.resize(2 + rng() % (m.size() - 2));                 // truncated
          else if (m.size() > 2) for (int k = 0; k < 1 + (int)(rng() % 4); ++k) m[2 + rng() % (m.size() - 2)] ^= (uint8_t)(1u << (rng() % 8));
          const size_t cap = kind == 2 ? (raw.size() ? rng() % raw.size() : 0) : raw.size() + rng() % 64;
          uint8_t* out = (uint8_t*)malloc(cap ? cap : 1);
          const long long got = dlpdf_fast_inflate(m.data(), m.size(), out, cap);
          if (got > (long long)cap) { fprintf(stderr, "inflate reported more bytes than the buffer holds\n"); exit(6); }
          refused += got < 0;
          ++total;
          free(out);
        }
      }
  printf("  inflate: round trips exact; %lld of %lld damaged / short-buffer streams refused, none overran\n", refused, total);
}

extern "C" int dlrt_layout_order(const float* boxes, int n, const float* figs, int nf, float page_h, int* order_out,
                                 int* block_out, int* kind_out);

// Layout analysis on pages it was not written for: text columns, grids of cells with wrapped lines and holes, and
// boxes that make no sense (empty, inverted, huge, all equal, thousands of them).  Whatever comes in, the answer is a
// permutation of the input with a block and a kind for every entry.
static void layout_stress() {
  std::mt19937 rng(11);
  auto uni = [&](float a, float b) { return a + (b - a) * (float)(rng() % 100000) / 100000.f; };
  long long pages = 0, cells = 0;
  for (int it = 0; it < 600; ++it) {
    std::vector<float> bx, fg;
    auto put = [&](float x0, float y0, float x1, float y1) { bx.push_back(x0); bx.push_back(y0); bx.push_back(x1); bx.push_back(y1); };
    const int mode = it % 6;
    const float h = uni(20, 70);
    if (mode == 0 || mode == 1) {                       // one or two text columns
      const int ncol = 1 + mode;
      for (int c = 0; c < ncol; ++c)
        for (float y = 200; y < 3000; y += h * uni(1.3f, 1.8f))
          put(200 + c * 1100, y, 200 + c * 1100 + uni(300, 1000), y + h);
    } else if (mode == 2 || mode == 3) {                // a table under a paragraph: wrapped cells, missing cells
      float y = 200;
      for (int k = 0; k < 4; ++k, y += h * 1.5f) put(200, y, 2200, y + h);
      const int cols = 2 + (int)(rng() % 4), rows = 3 + (int)(rng() % 9);
      const float cw = 2000.f / cols, pitch = h * uni(1.3f, 2.6f);
      for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < cols; ++c)
          if (mode == 2 || rng() % 7) put(200 + c * cw + 10, y, 200 + c * cw + uni(40, cw - 30), y + h);
        if (rng() % 3 == 0) { y += h * 1.25f; put(200 + 10, y, 200 + uni(40, cw - 30), y + h); }
        y += pitch;
      }
      for (int k = 0; k < 3; ++k, y += h * 1.5f) put(200, y + h, 2200, y + 2 * h);
    } else if (mode == 4) {                             // nonsense
      const int n = (int)(rng() % 40);
      for (int k = 0; k < n; ++k) {
        const int w = (int)(rng() % 5);
        if (w == 0) put(500, 500, 500, 500);
        else if (w == 1) put(uni(0, 2000), uni(0, 3000), uni(0, 2000), uni(0, 3000));      // possibly inverted
        else if (w == 2) put(-1e6f, -1e6f, 1e6f, 1e6f);
        else if (w == 3) put(100, 100, 900, 140);                                          // many equal boxes
        else put(uni(0, 2400), uni(0, 3400), uni(0, 2400) + 1e-3f, uni(0, 3400) + 1e-3f);
      }
    } else {                                            // many boxes
      const int n = 2000 + (int)(rng() % 2000);
      for (int k = 0; k < n; ++k) { const float x = uni(0, 2300), y = uni(0, 3400); put(x, y, x + uni(5, 300), y + uni(5, 60)); }
    }
    const int nf = (int)(rng() % 3);
    for (int k = 0; k < nf; ++k) { const float x = uni(100, 1500), y = uni(100, 2500); fg.insert(fg.end(), {x, y, x + uni(50, 800), y + uni(50, 800)}); }
    const int n = (int)bx.size() / 4;
    std::vector<int> order(n + nf + 1, -7), block(n + nf + 1, -7), kind(n + nf + 1, -7);
    const int got = dlrt_layout_order(bx.data(), n, nf ? fg.data() : nullptr, nf, 3508.f, order.data(), block.data(), kind.data());
    if (got != n + nf) { fprintf(stderr, "layout: %d entries for %d boxes + %d figures\n", got, n, nf); exit(3); }
    std::vector<char> seen(n + nf, 0);
    for (int k = 0; k < got; ++k) {
      if (order[k] < 0 || order[k] >= n + nf || seen[order[k]]) { fprintf(stderr, "layout: order is no permutation\n"); exit(3); }
      seen[order[k]] = 1;
      if (kind[k] < 0 || kind[k] > 2 || block[k] < 0) { fprintf(stderr, "layout: bad kind / block\n"); exit(3); }
      cells += kind[k] == 2;
    }
    if (order[got] != -7) { fprintf(stderr, "layout: wrote past the end\n"); exit(3); }
    ++pages;
  }
  printf("  layout: %lld pages ordered, %lld table cells, every answer a permutation\n", pages, cells);
}

int main(int argc, char** argv) {
  int iters = 1500;
  for (int i = 1; i < argc; ++i) {
    if (!strncmp(argv[i], "--iters=", 8)) { iters = atoi(argv[i] + 8); continue; }
    std::vector<uint8_t> pdf = read_file(argv[i]);
    if (pdf.empty()) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
    long long chars = 0;
    int images = 0;
    const int n = walk(pdf, &chars, &images);
    printf("%s: %d pages, %lld text bytes, %d images\n", argv[i], n, chars, images);
    if (n <= 0) { fprintf(stderr, "well-formed seed did not open\n"); return 2; }
    fuzz_pdf(pdf, iters, 1234u + (uint32_t)i);
  }
  jpeg_crafted();
  inflate_stress();
  batcher_stress();
  bpe_stress();
  layout_stress();
  printf("sanitize: OK\n");
  return 0;
}
