"""Native PDF reader: structure variants, text operators, encodings, images, classifier, JPEG scan."""
import io
import zlib

import numpy as np
import pytest

from distributed_llms_amd.pdf import reader, synth

TEXT = [["Hello world, this is page one of a born-digital document with plenty of text.",
         "Second line (with parens) and a \\ backslash to exercise string escapes.", "",
         "Another paragraph begins here, after a vertical gap, and keeps going for a while."],
        ["Page two is short but still text."]]


@pytest.mark.parametrize("xref_stream", [False, True])
@pytest.mark.parametrize("tj", [False, True])
@pytest.mark.parametrize("compress", [False, True])
def test_text_extraction_variants(xref_stream, tj, compress):
    pdf = synth.make_text_pdf(TEXT, xref_stream=xref_stream, use_tj_arrays=tj, compress=compress)
    with reader.PdfDocument(pdf) as d:
        assert d.page_count == 2
        t0 = d.page_text(0)
        assert t0.split("\n\n") == ["\n".join(TEXT[0][:2]), TEXT[0][3]]      # lines, then a paragraph break
        assert d.page_text(1) == TEXT[1][0]
        c = d.classify()
        assert c.is_text and c.pages_image_dominant == 0 and c.garbled_ratio == 0
        pi = d.page_info(0)
        assert abs(pi.width_pt - 595.28) < 0.01 and pi.n_images == 0 and pi.chars_visible > 150


def _raw_pdf(objs, root=1):
    out = io.BytesIO()
    out.write(b"%PDF-1.4\n")
    offs = []
    for i, b in enumerate(objs, 1):
        offs.append(out.tell())
        out.write(b"%d 0 obj\n" % i + b + b"\nendobj\n")
    x = out.tell()
    out.write(b"xref\n0 %d\n0000000000 65535 f \n" % (len(objs) + 1))
    for o in offs:
        out.write(b"%010d 00000 n \n" % o)
    out.write(b"trailer\n<< /Size %d /Root %d 0 R >>\nstartxref\n%d\n%%%%EOF\n" % (len(objs) + 1, root, x))
    return out.getvalue()


def _stream(d, data):
    return b"<< " + d + b" /Length %d >>\nstream\n" % len(data) + data + b"\nendstream"


def test_encodings_tounicode_hex_strings_and_positioning():
    cmap = (b"/CIDInit /ProcSet findresource begin 12 dict begin begincmap\n1 begincodespacerange <0000> <FFFF> endcodespacerange\n"
            b"2 beginbfchar <0003> <0048> <0004> <00E9> endbfchar\n1 beginbfrange <0010> <0012> <006A> endbfrange\nendcmap end end")
    content = (b"BT /F1 12 Tf 72 700 Td (caf\\351 \\223quoted\\224) Tj ET\n"          # WinAnsi octal escapes
               b"BT /F2 12 Tf 72 680 Td <00030004001000110012> Tj ET\n"                 # Identity-H + ToUnicode
               b"BT /F3 12 Tf 72 660 Td (ab) Tj ET\n"                                   # Differences -> glyph names
               b"BT /F1 12 Tf 1 0 0 1 72 640 Tm (left) Tj 200 0 Td (right) Tj ET\n"     # same line, big gap -> space
               b"q 2 0 0 2 0 0 cm BT /F1 6 Tf 36 310 Td (scaled) Tj ET Q\n"
               b"BT /F1 12 Tf 3 Tr 72 600 Td (hidden layer) Tj ET")
    objs = [
        b"<< /Type /Catalog /Pages 2 0 R >>",
        b"<< /Type /Pages /Count 1 /Kids [3 0 R] >>",
        b"<< /Type /Page /Parent 2 0 R /MediaBox [0 0 612 792] /Contents 4 0 R /Resources << /Font << /F1 5 0 R /F2 6 0 R /F3 9 0 R >> >> >>",
        _stream(b"", content),
        b"<< /Type /Font /Subtype /Type1 /BaseFont /Helvetica /Encoding /WinAnsiEncoding >>",
        b"<< /Type /Font /Subtype /Type0 /BaseFont /X /Encoding /Identity-H /DescendantFonts [7 0 R] /ToUnicode 8 0 R >>",
        b"<< /Type /Font /Subtype /CIDFontType2 /BaseFont /X /DW 600 >>",
        _stream(b"", cmap),
        b"<< /Type /Font /Subtype /Type1 /BaseFont /Foo /Encoding << /BaseEncoding /WinAnsiEncoding /Differences [97 /Euro /fi] >> >>",
    ]
    with reader.PdfDocument(_raw_pdf(objs)) as d:
        lines = d.page_text(0).replace("\n\n", "\n").split("\n")
        assert lines[0] == "café “quoted”"
        assert lines[1] == "Héjkl"
        assert lines[2] == "€ﬁ"
        assert lines[3] == "left right"
        assert lines[4] == "scaled"
        assert "hidden layer" not in d.page_text(0) and "hidden layer" in d.page_text(0, include_invisible=True)
        pi = d.page_info(0)
        assert pi.chars_invisible == len("hiddenlayer") and pi.unmapped == 0


def test_object_streams_incremental_update_and_rebuild():
    # PDF 1.5 object stream holding the catalog and page tree
    inner = [b"<< /Type /Catalog /Pages 2 0 R >>", b"<< /Type /Pages /Count 1 /Kids [3 0 R] >>"]
    hdr, body = b"", b""
    for k, o in enumerate(inner):
        hdr += b"%d %d " % (k + 1, len(body))
        body += o + b"\n"
    objstm_data = zlib.compress(hdr + body)
    objstm = _stream(b"/Type /ObjStm /N 2 /First %d /Filter /FlateDecode" % len(hdr), objstm_data)
    page = b"<< /Type /Page /Parent 2 0 R /MediaBox [0 0 200 200] /Contents 4 0 R /Resources << /Font << /F1 5 0 R >> >> >>"
    content = _stream(b"", b"BT /F1 10 Tf 10 100 Td (from object stream) Tj ET")
    font = b"<< /Type /Font /Subtype /Type1 /BaseFont /Courier >>"
    out = io.BytesIO()
    out.write(b"%PDF-1.5\n")
    offs = {}
    for num, b in ((3, page), (4, content), (5, font), (6, objstm)):
        offs[num] = out.tell()
        out.write(b"%d 0 obj\n" % num + b + b"\nendobj\n")
    xpos = out.tell()
    rows = bytes([0, 0, 0, 0, 0]) + bytes([2, 0, 0, 6, 0]) + bytes([2, 0, 0, 6, 1])
    for num in (3, 4, 5, 6):
        rows += bytes([1]) + offs[num].to_bytes(3, "big") + bytes([0])
    rows += bytes([1]) + xpos.to_bytes(3, "big") + bytes([0])
    out.write(b"7 0 obj\n" + _stream(b"/Type /XRef /Size 8 /W [1 3 1] /Root 1 0 R", rows) + b"\nendobj\n")
    out.write(b"startxref\n%d\n%%%%EOF\n" % xpos)
    pdf = out.getvalue()
    with reader.PdfDocument(pdf) as d:
        assert d.page_count == 1 and d.page_text(0) == "from object stream" and not d.was_rebuilt
    # incremental update: a newer revision of the content stream appended with /Prev
    base = synth.make_text_pdf([["old text that will be replaced by an incremental update"]], compress=False)
    sx = int(base[base.rindex(b"startxref") + 9:].split()[0])
    new_obj = _stream(b"", b"BT /F1 11 Tf 72 770 Td (new text wins) Tj ET")
    upd = io.BytesIO()
    upd.write(base)
    pos = upd.tell()
    upd.write(b"4 0 obj\n" + new_obj + b"\nendobj\n")
    xp = upd.tell()
    upd.write(b"xref\n4 1\n%010d 00000 n \ntrailer\n<< /Size 6 /Root 1 0 R /Prev %d >>\nstartxref\n%d\n%%%%EOF\n" % (pos, sx, xp))
    with reader.PdfDocument(upd.getvalue()) as d:
        assert d.page_text(0) == "new text wins"
    # damaged xref offsets -> full rebuild by scanning for objects
    broken = base.replace(b"startxref\n%d" % sx, b"startxref\n%d" % (sx + 7))
    with reader.PdfDocument(broken) as d:
        assert d.was_rebuilt and "old text" in d.page_text(0)
    with pytest.raises(reader.PdfError):
        reader.PdfDocument(b"%PDF-1.4\nnothing useful")
    with pytest.raises(reader.PdfError):
        reader.PdfDocument(b"plain text")


def test_scanned_pages_images_and_classifier():
    pg, _ = synth.render_page(2, size=(320, 452), font_px=11, margin=24)
    rgb = np.stack([pg, pg // 2, 255 - pg], axis=-1)
    for codec in ("flate", "jpeg"):
        pdf = synth.make_scanned_pdf([pg, rgb], codec=codec, dpi=40, restart_blocks=10)
        with reader.PdfDocument(pdf) as d:
            c = d.classify()
            assert not c.is_text and c.pages_image_dominant == 2 and c.avg_coverage > 0.99
            im0, im1 = d.main_image(0), d.main_image(1)
            assert (im0.width, im0.height, im0.out_channels) == (320, 452, 1) and im1.out_channels == 3
            if codec == "flate":
                assert im0.codec == reader.CODEC_RAW
                assert np.array_equal(d.decode_image(im0), pg) and np.array_equal(d.decode_image(im1), rgb)
                buf = np.zeros(320 * 452 + 64, dtype=np.uint8)
                assert np.array_equal(d.decode_image(im0, buf), pg)            # decode into caller memory
            else:
                assert im0.codec == reader.CODEC_JPEG and im0.stream_off > 0
                sc = reader.scan_jpeg(d.jpeg_bytes(im0))
                assert (sc.info.width, sc.info.height, sc.info.ncomp) == (320, 452, 1)
                assert sc.info.restart_interval == 10 and sc.gpu_decodable
                assert sc.info.n_segments == -(-(sc.info.mcus_x * sc.info.mcus_y) // 10)
                assert sc.seg_offsets[0] == sc.info.scan_off and np.all(np.diff(sc.seg_offsets.astype(np.int64)) > 2)


def test_image_formats_bilevel_indexed_predictor():
    W, H = 37, 9
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2, size=(H, W), dtype=np.uint8)
    packed = np.packbits(bits, axis=1).tobytes()
    idx = rng.integers(0, 4, size=(H, W), dtype=np.uint8)
    palette = bytes([0, 0, 0, 255, 0, 0, 0, 255, 0, 255, 255, 255])
    gray = rng.integers(0, 255, size=(H, W), dtype=np.uint8)
    # PNG "Sub" predictor rows
    pred = b"".join(b"\x01" + np.diff(np.concatenate([[0], r.astype(np.int16)])).astype(np.uint8).tobytes() for r in gray)

    def page_with(img_dict, data):
        objs = [b"<< /Type /Catalog /Pages 2 0 R >>", b"<< /Type /Pages /Count 1 /Kids [3 0 R] >>",
                b"<< /Type /Page /Parent 2 0 R /MediaBox [0 0 100 100] /Contents 4 0 R /Resources << /XObject << /I 5 0 R >> >> >>",
                _stream(b"", b"q 100 0 0 100 0 0 cm /I Do Q"),
                _stream(b"/Type /XObject /Subtype /Image /Width %d /Height %d " % (W, H) + img_dict, data)]
        return _raw_pdf(objs)

    with reader.PdfDocument(page_with(b"/ColorSpace /DeviceGray /BitsPerComponent 1", packed)) as d:
        assert np.array_equal(d.decode_image(d.main_image(0)), bits * 255)
    with reader.PdfDocument(page_with(b"/ImageMask true /BitsPerComponent 1", packed)) as d:
        assert np.array_equal(d.decode_image(d.main_image(0)), (1 - bits) * 255)   # stencil: 0 paints ink
    with reader.PdfDocument(page_with(b"/ColorSpace [/Indexed /DeviceRGB 3 <%s>] /BitsPerComponent 8" % palette.hex().encode(),
                                      idx.tobytes())) as d:
        out = d.decode_image(d.main_image(0))
        assert out.shape == (H, W, 3) and np.array_equal(out, np.frombuffer(palette, np.uint8).reshape(4, 3)[idx])
    dp = b"/ColorSpace /DeviceGray /BitsPerComponent 8 /Filter /FlateDecode /DecodeParms << /Predictor 11 /Columns %d >>" % W
    with reader.PdfDocument(page_with(dp, zlib.compress(pred))) as d:
        assert np.array_equal(d.decode_image(d.main_image(0)), gray)
    with reader.PdfDocument(page_with(b"/ColorSpace /DeviceGray /BitsPerComponent 1 /Filter /JBIG2Decode", b"xx")) as d:
        assert d.main_image(0).codec == reader.CODEC_UNSUPPORTED


def test_short_born_digital_document_is_text_not_ocr():
    with reader.PdfDocument(synth.make_text_pdf([["Tiny note."]])) as d:
        assert d.classify().is_text


@pytest.mark.parametrize("rotate", [90, 180, 270])
def test_rotated_scans_are_read_upright(rotate):
    """A page stored sideways with /Rotate (a scanner's landscape feed) must give the same lines as the upright
    scan: the reader reports the angle, the pipeline turns the raster before detection and cropping."""
    import torch

    from distributed_llms_amd.runtime.document import DocumentPipeline
    from distributed_llms_amd.runtime.workers import build_engine

    rendered = [synth.render_page(40 + i, size=(248, 350), font_px=10, margin=16) for i in range(2)]
    pages = [p for p, _ in rendered]
    priors = [t.boxes for _, t in rendered]
    eng = build_engine("cpu", {"model": "tiny", "cfg": dict(det_limit_side=160, rec_max_width=256, rec_width_quantum=64,
                                                             max_batch_pages=2)})
    pipe = DocumentPipeline(eng, decode_threads=2)
    up = pipe.process_pdf(synth.make_scanned_pdf(pages, codec="flate", dpi=30), "up", layout_priors=priors)
    pdf = synth.make_scanned_pdf(pages, codec="flate", dpi=30, rotate=rotate)
    with reader.PdfDocument(pdf) as doc:
        assert doc.page_info(0).rotate == rotate
    rot = pipe.process_pdf(pdf, "rot", layout_priors=priors)
    assert [(p.width, p.height) for p in rot.pages] == [(p.width, p.height) for p in up.pages]
    assert rot.text == up.text and len(rot.pages[0].lines) == len(priors[0])


def _bilevel(seed, H, W):
    pg, _ = synth.render_page(seed, size=(W, H), font_px=14, margin=20, noise=0.0)
    return pg


@pytest.mark.parametrize("hw", [(350, 248), (97, 131), (64, 2600)])
def test_ccitt_group4_scans_decode_exactly(hw):
    """CCITTFaxDecode /K -1 (what office scanners write for black-and-white pages): the native decoder must
    reproduce the bilevel page bit for bit (encoder: PIL/libtiff)."""
    H, W = hw
    rng = np.random.default_rng(H * 7 + W)
    pages = [_bilevel(3, H, W), ((rng.random((H, W)) > 0.35) * 255).astype(np.uint8),          # text; dense noise
             np.full((H, W), 255, np.uint8), np.zeros((H, W), np.uint8)]                         # all white; all black
    pages[1][:, : W // 3] = 255                                                                   # long white runs too
    pdf = synth.make_scanned_pdf(pages, codec="ccitt", dpi=72)
    with reader.PdfDocument(pdf) as doc:
        assert doc.page_count == len(pages)
        for i, pg in enumerate(pages):
            im = doc.main_image(i)
            assert im is not None and im.codec == reader.CODEC_RAW and (im.width, im.height) == (W, H)
            got = doc.decode_image(im)
            want = ((pg >= 128) * 255).astype(np.uint8)
            assert got.shape == want.shape
            assert np.array_equal(got, want), f"page {i}: {(got != want).sum()} pixels differ"
            # the same page as packed bits (what goes to the GPU: 8 pixels per byte, set bit = white)
            bits = doc.decode_image_bits(im)
            assert bits is not None and bits.shape == (H, (W + 7) // 8)
            assert np.array_equal(np.unpackbits(bits, axis=1)[:, :W] * 255, want)
        assert not doc.classify().is_text
    # a gray page is no bilevel image: the packed path declines
    with reader.PdfDocument(synth.make_scanned_pdf([pages[0]], codec="flate", dpi=72)) as doc:
        assert doc.decode_image_bits(doc.main_image(0)) is None


def test_ccitt_damaged_data_never_crashes():
    pdf = bytearray(synth.make_scanned_pdf([_bilevel(5, 200, 160)], codec="ccitt", dpi=72))
    start = pdf.index(b"stream\n") + 7
    rng = np.random.default_rng(0)
    for trial in range(40):
        bad = bytearray(pdf)
        for _ in range(1 + trial % 5):
            bad[start + int(rng.integers(0, 300))] = int(rng.integers(0, 256))
        try:
            with reader.PdfDocument(bytes(bad)) as doc:
                im = doc.main_image(0)
                if im is not None and im.codec == reader.CODEC_RAW:
                    out = doc.decode_image(im)
                    assert out.shape == (200, 160)
        except reader.PdfError:
            pass


def test_process_pdfs_streams_documents_and_matches_one_at_a_time():
    """DocumentPipeline.process_pdfs: several documents through one engine pipeline = the same results, input order."""
    import torch

    from distributed_llms_amd.models.crnn import CRNN
    from distributed_llms_amd.models.dbnet import DBNet
    from distributed_llms_amd.pdf import synth
    from distributed_llms_amd.runtime.document import DocumentPipeline
    from distributed_llms_amd.runtime.engine import EngineConfig, OcrEngine

    torch.manual_seed(0)
    cfg = EngineConfig(max_batch_pages=2, use_hip_graphs=False)
    cfg.det_limit_side, cfg.rec_max_width, cfg.rec_width_quantum = 160, 256, 64
    eng = OcrEngine(torch.device("cpu"), DBNet(widths=(16, 32, 32, 64), stem=8, inner=32),
                    CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64)), cfg, seed=1)
    pipe = DocumentPipeline(eng)
    docs = []
    for k, npages in enumerate((3, 1, 2)):
        rendered = [synth.render_page(10 * k + i, size=(248, 350), font_px=10, margin=16) for i in range(npages)]
        pdf = synth.make_scanned_pdf([r[0] for r in rendered], codec="flate")
        docs.append((pdf, f"doc{k}", [r[1].boxes for r in rendered]))
    docs.append((synth.make_text_pdf(["alpha beta", "gamma"]), "txt", None))
    many = pipe.process_pdfs(docs)
    assert [d.document_id for d in many] == ["doc0", "doc1", "doc2", "txt"]
    for (pdf, did, pri), got in zip(docs, many):
        want = pipe.process_pdf(pdf, did, layout_priors=pri)
        assert got.mode == want.mode and got.text == want.text
        assert [p.page_index for p in got.pages] == [p.page_index for p in want.pages]
        assert all(p.doc_id == did for p in got.pages)
    pipe.close()


def _jpeg_bytes(arr, **kw):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(arr).save(b, format="JPEG", **kw)
    return np.frombuffer(b.getvalue(), dtype=np.uint8)


@pytest.mark.parametrize("case", [
    dict(kind="text", size=(620, 877), quality=85),
    dict(kind="text", size=(620, 877), quality=95, optimize=True),
    dict(kind="noise", size=(312, 200), quality=90),          # long codes, stuffed 0xFF bytes at subsequence edges
    dict(kind="rgb", size=(320, 240), quality=85),            # 4:2:0: six blocks per MCU, two table pairs
    dict(kind="rgb", size=(320, 240), quality=85, subsampling=0),
    dict(kind="blank", size=(2480, 3508), quality=85),        # never resynchronises: bounded rounds, repeated seam passes
])
def test_restartless_jpeg_self_synchronising_decode_matches_libjpeg(case):
    """The pipeline the GPU runs for scans without restart markers (guessed-state subsequence decode, merge rounds,
    seam passes, two prefix sums), executed serially on the host from the same source."""
    from PIL import Image

    case = dict(case)
    kind = case.pop("kind")
    W, H = case.pop("size")
    rng = np.random.default_rng(1)
    if kind == "text":
        img, _ = synth.render_page(4, size=(W, H), font_px=14, margin=20, noise=3.0)
    elif kind == "noise":
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
    elif kind == "blank":
        img = np.full((H, W), 255, dtype=np.uint8)
        img[40:60, 100:900] = 0
    else:
        g, _ = synth.render_page(4, size=(W, H), font_px=12, margin=10)
        img = np.stack([g, g // 2 + 60, 255 - g], axis=-1)
    st = _jpeg_bytes(img, **case)
    sc = reader.scan_jpeg(st)
    assert sc.gpu_decodable and sc.self_synchronising and sc.info.n_segments == 1
    got, stats = reader.decode_jpeg_sync_host(st)
    assert stats.blocks_decoded == sc.info.mcus_x * sc.info.mcus_y * sc.info.blocks_per_mcu
    im = Image.open(io.BytesIO(st.tobytes()))
    im.draft("L", im.size)
    ref = np.asarray(im).astype(np.int32)
    d = np.abs(got.astype(np.int32) - ref)
    assert d.max() <= 3 and d.mean() < 0.5, (d.max(), d.mean())
    if kind == "blank":
        assert stats.seam_passes > 1 and stats.max_rounds == 255
    elif kind == "text":
        # every lane falls into step within a few subsequences: ~2-3 decodes of each bit in total
        assert stats.max_rounds <= 24 and stats.sub_decodes <= 4 * stats.n_sub


def test_jpeg_routing_by_restart_structure():
    img, _ = synth.render_page(4, size=(1240, 1754), font_px=22, margin=40)
    a = reader.scan_jpeg(_jpeg_bytes(img, quality=80))
    b = reader.scan_jpeg(_jpeg_bytes(img, quality=80, restart_marker_blocks=62))
    c = reader.scan_jpeg(_jpeg_bytes(img, quality=80, restart_marker_rows=40))      # 6 long segments: host
    d = reader.scan_jpeg(_jpeg_bytes(img, quality=80, progressive=True))
    assert (a.info.supported, b.info.supported, c.info.supported, d.info.supported) == (2, 1, 0, 0)


def test_in_repo_inflater_matches_zlib_and_refuses_what_it_cannot_finish():
    """csrc/pdf/fast_inflate.h: exact round trips over stored / fixed / dynamic blocks, run-length and Huffman-only
    streams and sizes around the fast loop's margins; a stream that is cut short or does not fit the buffer is refused
    (the library then falls back to zlib), never decoded past the buffer."""
    import zlib

    rng = np.random.default_rng(0)
    page = np.ascontiguousarray(synth.render_page(3, size=(620, 877), font_px=14, margin=40)[0]).tobytes()
    cases = [b"", b"a", b"abc" * 1000, bytes(rng.integers(0, 256, 100000, dtype=np.uint8)), b"\0" * 300000,
             bytes(rng.integers(0, 4, 200000, dtype=np.uint8)), page]
    cases += [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (7, 8, 9, 16, 17, 319, 320, 321, 400)]
    for raw in cases:
        for level in (0, 1, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                c = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
                z = c.compress(raw) + c.flush()
                for cap in (len(raw), len(raw) + 5, len(raw) + 4096):
                    got = reader.fast_inflate(z, cap)
                    assert got is not None and bytes(got) == raw, (len(raw), level, strategy, cap)
                if raw:
                    assert reader.fast_inflate(z, len(raw) - 1) is None
    z = zlib.compress(page, 6)
    for cut in (1, 2, 5, 100, len(z) // 2, len(z) - 5):
        got = reader.fast_inflate(z[:cut], len(page))
        assert got is None or bytes(got) == page[: len(got)]
    assert reader.fast_inflate(b"\x78\x9d" + z[2:], len(page)) is None            # header check bits do not add up
    assert reader.fast_inflate(b"\x78\xbb" + z[2:], len(page)) is None            # preset dictionary: not ours


def test_flate_scan_with_png_predictors_decodes_like_the_plain_one():
    """A Flate image whose rows carry PNG predictor bytes (what image tools write) gives the same raster as the
    unpredicted stream: the plain stream is inflated straight into the caller's buffer, the predicted one goes through
    the filter chain.  Gray and RGB, every predictor type."""
    gray = synth.render_page(5, size=(310, 438), font_px=12, margin=16)[0]
    rgb = np.stack([gray, 255 - gray, (gray.astype(np.int32) * 3 // 4).astype(np.uint8)], -1)
    for page in (gray, rgb):
        for predictor in (1, 11, 12, 13, 14, 15):
            doc = reader.PdfDocument(synth.make_scanned_pdf([page], codec="flate", dpi=40, predictor=predictor))
            got = doc.decode_image(doc.main_image(0))
            assert np.array_equal(got, page), (page.ndim, predictor)
            doc.close()


def test_synthetic_pages_with_formulas_inside_lines_keep_their_truth_consistent():
    """``render_layout_page(inline=...)``: the truth text of a line with formulas is its parts joined by spaces, formulas
    as $...$; every formula box lies inside its line's box, left to right as the parts say; pages without ``inline``
    are drawn exactly as before."""
    from distributed_llms_amd.pdf import synth

    pg, tr = synth.render_layout_page(77, size=(1240, 1754), layout="single", inline=0.6, noise=0.0)
    assert len(tr.line_parts) >= 5 and tr.inline.shape[0] == len(tr.inline_texts) == len(tr.inline_line) >= 5
    for li, parts in tr.line_parts.items():
        assert tr.texts[li] == " ".join(t if k == "t" else "$" + t + "$" for k, _, _, t in parts)
        xs = [x for _, x0, x1, _ in parts for x in (x0, x1)]
        assert xs == sorted(xs)
        assert any(k == "t" for k, *_ in parts) and any(k == "m" for k, *_ in parts)
    for box, li, latex in zip(tr.inline, tr.inline_line, tr.inline_texts):
        lb = tr.boxes[li]
        assert lb[0] - 1 <= box[0] < box[2] <= lb[2] + 1 and lb[1] - 1 <= box[1] < box[3] <= lb[3] + 1
        assert "$" + latex + "$" in tr.texts[li]
    # the formula is set on the text baseline: ink of the formula and of the words share rows
    li = tr.inline_line[0]
    fb = tr.inline[0].astype(int)
    rows_formula = np.nonzero((pg[fb[1]:fb[3], fb[0]:fb[2]] < 128).any(1))[0] + fb[1]
    lb = tr.boxes[li].astype(int)
    assert rows_formula.size and lb[1] <= rows_formula.min() and rows_formula.max() <= lb[3]
    a, _ = synth.render_layout_page(78, size=(620, 877), layout="two_col", formulas=0.4)
    b, tb = synth.render_layout_page(78, size=(620, 877), layout="two_col", formulas=0.4, inline=0.0)
    assert np.array_equal(a, b) and not tb.line_parts
    r = synth.render_formula_baseline("x_{i}^{2} + \\alpha", 40)
    assert r is not None and 0 < r[1] <= r[0].shape[0] + 2
