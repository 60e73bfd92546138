"""Synthetic documents: rendered "scanned" pages, text lines for recognizer training, and PDF files.

There is no dataset and no network in the build environment, and the reference
ships no fixtures, so every workload is generated here:

* ``render_page``      - one A4 page at 300 DPI (2480 x 3508, 8-bit gray) with
  paragraphs of text, plus the ground-truth line boxes and strings;
* ``render_line``      - one text-line crop for CTC training;
* ``make_scanned_pdf`` - a PDF whose pages are one full-page image XObject each
  (DCTDecode with restart markers, or FlateDecode) - the shape a scanner emits
  and the workload of the scanned-document benchmarks;
* ``make_text_pdf``    - a born-digital PDF with Type-1 Helvetica text
  (exercises the C++ reader's text operators).

PIL (with FreeType and libjpeg-turbo) is the only dependency.
"""
from __future__ import annotations

import io
import zlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
from PIL import Image, ImageDraw, ImageFont

A4_300DPI = (2480, 3508)  # width, height in pixels

_WORDS = (
    "the of and to in is that for it as was with be by on not he this are or his from at which but have an "
    "had they you were their one all we can her has there been if more when will would who so no out up into "
    "page batch kernel wave tensor matrix memory stream graph device vector scalar cache tile fused compute "
    "document service status result extract upload health queue worker shard gather reduce scatter overlap "
    "Invoice Total Amount Date Number Reference Section Figure Table Chapter Summary Report Annex Appendix "
    "2024 2025 1,250.00 37% 4.5 No. 12 (a) [3] A-17 x=3 y<9 #42 $99 & / : ; ? !"
).split()

# Western European vocabulary (German, French, Spanish, Italian, Portuguese, Scandinavian) and the signs that come
# with it: every character of models/crnn.py::LATIN_EXTRA occurs at least once
_WORDS_I18N = (
    "über für können müssen Straße Größe Änderung Österreich schön während möglich natürlich Geschäft Prüfung Lösung "
    "Verfügung zurück fünf zwölf Maß weiß heißt Ägypten Ärzte Ölpreis Übung Fuß Mädchen Bücher "
    "été être déjà où à très après même français façon reçu élève hôtel garçon forêt naïve Noël cœur œuvre sœur août "
    "coût crème tête École État Île Œuvre Ça Èze Êtes aiguë maïs où dû sûr L'Haÿ "
    "año señor niño mañana español también después más está aquí cómo qué día información corazón ¿Cómo? ¡Hola! "
    "pingüino Ñandú Óscar Úbeda Ímpar Ávila Él "
    "perché più città è così università qualità già però lunedì caffè Àsia Ìsola Òpera Ùltimo "
    "não são ação coração informação você português maçã pão irmã órgão avó avô Ã Õ Â Ê Î Ô Û Ë Ï Ý ý "
    "år också är på både språk smör gå sjö kött blå Århus København fjørd smørrebrød Ørsted Ælfred æble Åland Ýmir "
    "île âme põe âge maître dîner lições "
    "12,50 € £30 §4 25 °C «oui» «non» 3 € § 7 90° £ 1.200,00 €"
).split()

_FONT_CACHE = {}


def font(size: int, bold: bool = False) -> ImageFont.FreeTypeFont:
    key = (size, bold)
    f = _FONT_CACHE.get(key)
    if f is None:
        try:
            name = "DejaVuSans-Bold.ttf" if bold else "DejaVuSans.ttf"
            f = ImageFont.truetype(name, size)
        except OSError:
            f = ImageFont.load_default(size=size)
        _FONT_CACHE[key] = f
    return f


def random_text(rng: np.random.Generator, max_chars: int, min_words: int = 1, vocabulary: Optional[Sequence[str]] = None) -> str:
    words: List[str] = []
    n = 0
    _WORDS = vocabulary or globals()["_WORDS"]
    while True:
        w = _WORDS[int(rng.integers(len(_WORDS)))]
        if rng.random() < 0.08:
            w = w.capitalize()
        if n + len(w) + (1 if words else 0) > max_chars and len(words) >= min_words:
            break
        words.append(w)
        n += len(w) + 1
        if n >= max_chars:
            break
    return " ".join(words)[:max_chars].strip()


@dataclass
class PageTruth:
    boxes: np.ndarray            # [n, 4] float32 x0, y0, x1, y1 (pixels, reading order; upright page frame)
    texts: List[str] = field(default_factory=list)
    figures: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))   # figure boxes, same frame
    blocks: List[int] = field(default_factory=list)      # text block id per line (lines of a block form paragraphs)
    angle: float = 0.0           # degrees the finished page was turned counter-clockwise about its centre (skew)
    layout: str = "single"
    font_names: List[str] = field(default_factory=list)
    formulas: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))  # display formulas, same frame
    formula_texts: List[str] = field(default_factory=list)     # their LaTeX source
    formula_pos: List[int] = field(default_factory=list)       # number of text lines that precede each formula
    # formulas inside text lines: span boxes (same frame), LaTeX source, index of the line in ``texts`` (whose text
    # carries the formula as ``$...$``), and per such line its parts left to right: ("t" | "m", x0, x1, text)
    inline: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))
    inline_texts: List[str] = field(default_factory=list)
    inline_line: List[int] = field(default_factory=list)
    line_parts: dict = field(default_factory=dict)

    def rotated_rects(self, size: Tuple[int, int]) -> np.ndarray:
        """Line boxes in the frame of the skewed image: [n, 5] cx, cy, w, h, angle (degrees, counter-clockwise)."""
        return rotate_boxes(self.boxes, self.angle, size)


def rotate_boxes(boxes: np.ndarray, angle: float, size: Tuple[int, int]) -> np.ndarray:
    """Upright boxes -> (cx, cy, w, h, angle) after the page was turned ``angle`` degrees counter-clockwise about its
    centre (image coordinates, y down)."""
    W, H = size
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    cx, cy = 0.5 * (b[:, 0] + b[:, 2]) - 0.5 * W, 0.5 * (b[:, 1] + b[:, 3]) - 0.5 * H
    a = np.deg2rad(angle)
    # counter-clockwise on screen with y pointing down: x' = x cos + y sin, y' = -x sin + y cos
    rx, ry = cx * np.cos(a) + cy * np.sin(a), -cx * np.sin(a) + cy * np.cos(a)
    out = np.stack([rx + 0.5 * W, ry + 0.5 * H, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1], np.full(b.shape[0], angle)], 1)
    return out.astype(np.float32)


def render_page(seed: int, size: Tuple[int, int] = A4_300DPI, font_px: int = 42, noise: float = 2.0,
                margin: int = 220) -> Tuple[np.ndarray, PageTruth]:
    """One synthetic scanned page: uint8 [H, W] (0 = ink, ~245 = paper) and its line truth."""
    W, H = size
    rng = np.random.default_rng(seed)
    img = Image.new("L", (W, H), 246)
    d = ImageDraw.Draw(img)
    f = font(font_px)
    fb = font(int(font_px * 1.35), bold=True)
    line_h = int(font_px * 1.55)
    y = margin
    boxes, texts = [], []
    avg_char = f.getlength("abcdefghij") / 10.0
    max_chars = max(8, int((W - 2 * margin) / avg_char))
    first = True
    while y + line_h < H - margin:
        if first or rng.random() < 0.07:
            # heading
            t = random_text(rng, max(8, max_chars // 3)).title()
            ff = fb
            first = False
            y += line_h // 2
        else:
            # paragraph line; the last line of a paragraph is short
            frac = 1.0 if rng.random() > 0.18 else float(rng.uniform(0.25, 0.8))
            t = random_text(rng, max(8, int(max_chars * frac)), min_words=2)
            ff = f
        if not t:
            continue
        x = margin + (0 if rng.random() > 0.1 else int(rng.integers(40, 160)))
        bb = d.textbbox((x, y), t, font=ff)
        if bb[2] > W - margin // 2:
            continue
        d.text((x, y), t, fill=int(rng.integers(10, 50)), font=ff)
        boxes.append([bb[0] - 4, bb[1] - 4, bb[2] + 4, bb[3] + 4])
        texts.append(t)
        y += line_h if ff is f else int(line_h * 1.5)
        if rng.random() < 0.12:
            y += line_h  # paragraph gap
    arr = np.asarray(img, dtype=np.uint8)
    if noise > 0:
        # scanner-like sensor noise on a coarse grid (keeps JPEG/Flate sizes realistic, cheap to make)
        nz = rng.normal(0.0, noise, size=(H // 4 + 1, W // 4 + 1)).astype(np.float32)
        nz = np.repeat(np.repeat(nz, 4, axis=0), 4, axis=1)[:H, :W]
        arr = np.clip(arr.astype(np.float32) + nz, 0, 255).astype(np.uint8)
    return arr, PageTruth(np.asarray(boxes, dtype=np.float32).reshape(-1, 4), texts)


# ------------------------------------------------------------------------------------------- fonts
_FONT_DIRS = ("/usr/share/fonts/truetype/dejavu",
              "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf",
              "/usr/local/lib/python3.10/dist-packages/kaleido/executable/xdg/fonts/truetype/lato")
# font files with full ASCII coverage that exist in the build image, by family
TRAIN_FONTS = ("DejaVuSans.ttf", "DejaVuSans-Bold.ttf", "DejaVuSans-Oblique.ttf", "DejaVuSerif.ttf", "DejaVuSerif-Bold.ttf",
               "DejaVuSerif-Italic.ttf", "DejaVuSansMono.ttf", "DejaVuSansMono-Bold.ttf", "STIXGeneral.ttf",
               "STIXGeneralBol.ttf", "STIXGeneralItalic.ttf", "cmr10.ttf", "cmtt10.ttf", "cmb10.ttf")
HELD_OUT_FONTS = ("Lato-Regular.ttf", "Lato-Bold.ttf", "Lato-Italic.ttf", "cmss10.ttf")   # never rendered in training
# Cyrillic and Greek exist in three families here (DejaVu, STIX, Lato).  Pages in those alphabets are trained on DejaVu
# and on Lato cuts that are NOT in ``HELD_OUT_FONTS`` (Latin training never draws any Lato file), and evaluated on the
# STIX family - never drawn in these alphabets - plus the three held-out Lato cuts.  Lato is in the training side
# because DejaVu's Greek is the odd one out (a round nu; nearly every text face draws nu pointed like a v).
SCRIPT_TRAIN_FONTS = ("DejaVuSans.ttf", "DejaVuSans-Bold.ttf", "DejaVuSans-Oblique.ttf", "DejaVuSans-BoldOblique.ttf",
                      "DejaVuSerif.ttf", "DejaVuSerif-Bold.ttf", "DejaVuSerif-Italic.ttf", "DejaVuSerif-BoldItalic.ttf",
                      "DejaVuSansMono.ttf", "DejaVuSansMono-Bold.ttf", "Lato-Light.ttf", "Lato-Medium.ttf",
                      "Lato-Semibold.ttf", "Lato-Heavy.ttf", "Lato-LightItalic.ttf", "Lato-MediumItalic.ttf",
                      "Lato-SemiboldItalic.ttf", "Lato-HeavyItalic.ttf")
SCRIPT_HELD_OUT_FONTS = ("STIXGeneral.ttf", "STIXGeneralBol.ttf", "STIXGeneralItalic.ttf", "STIXGeneralBolIta.ttf",
                         "Lato-Regular.ttf", "Lato-Bold.ttf", "Lato-Italic.ttf")
# the Computer Modern files carry TeX encodings: these code points show other glyphs there
_CM_UNSAFE = set('<>\\_{|}"~^`')
_FONT_PATHS = {}


def font_path(name: str) -> Optional[str]:
    import os

    if name in _FONT_PATHS:
        return _FONT_PATHS[name]
    found = None
    for d in _FONT_DIRS:
        p = os.path.join(d, name)
        if os.path.exists(p):
            found = p
            break
    _FONT_PATHS[name] = found
    return found


def available_fonts(names: Sequence[str]) -> List[str]:
    return [n for n in names if font_path(n) is not None]


def named_font(name: str, size: int) -> ImageFont.FreeTypeFont:
    key = (name, size)
    f = _FONT_CACHE.get(key)
    if f is None:
        p = font_path(name)
        f = ImageFont.truetype(p, size) if p else font(size)
        _FONT_CACHE[key] = f
    return f


_FONT_CMAPS = {}


def font_code_points(name: str) -> Optional[frozenset]:
    """Code points the font file maps to a glyph (None when the file or fontTools is missing: assume everything)."""
    if name not in _FONT_CMAPS:
        cps = None
        path = font_path(name)
        if path:
            try:
                from fontTools.ttLib import TTFont

                cps = frozenset(TTFont(path, lazy=True).getBestCmap().keys())
            except Exception:  # noqa: BLE001 - no fontTools / unreadable table: do not filter
                cps = None
        _FONT_CMAPS[name] = cps
    return _FONT_CMAPS[name]


def _font_safe(text: str, name: str) -> str:
    """``text`` without the characters ``name`` cannot draw (the truth is what is on the page)."""
    if name.startswith("cm"):
        text = "".join(c for c in text if c not in _CM_UNSAFE)
    if not text.isascii():
        cps = font_code_points(name)
        if cps is not None:
            text = "".join(c for c in text if ord(c) < 128 or ord(c) in cps)
        if name.startswith("cm"):
            text = "".join(c for c in text if ord(c) < 128)      # TeX encodings: only the ASCII range is what it says
        text = " ".join(text.split())
    return text.strip() if name.startswith("cm") else text


def _figure_patch(rng: np.random.Generator, w: int, h: int) -> np.ndarray:
    """Synthetic "photo / chart" content: a smooth gradient, a few soft blobs and bars, sensor noise."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    g = rng.uniform(60, 200) + rng.uniform(-60, 60) * xx / max(w, 1) + rng.uniform(-60, 60) * yy / max(h, 1)
    for _ in range(int(rng.integers(2, 6))):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(0.08, 0.3) * min(w, h)
        g += rng.uniform(-90, 90) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))
    if rng.random() < 0.5:                       # bar chart
        nb = int(rng.integers(3, 9))
        bw = w // (2 * nb + 1)
        for k in range(nb):
            bh = int(rng.uniform(0.15, 0.85) * h)
            g[h - bh : h, (2 * k + 1) * bw : (2 * k + 2) * bw] = rng.uniform(20, 120)
    g += rng.normal(0, 6, size=g.shape)
    return np.clip(g, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------------------- formulas
_F_VARS = list("abcdefghkmnpqrstuvwxyz") + list("ABCDEFGHKLMNPQRSTVXYZ")
_F_GREEK = ["\\alpha", "\\beta", "\\gamma", "\\delta", "\\epsilon", "\\theta", "\\lambda", "\\mu", "\\pi", "\\rho",
            "\\sigma", "\\phi", "\\omega", "\\Delta", "\\Omega", "\\Sigma"]
_F_OPS = ["+", "-", "=", "\\cdot", "\\times", "\\leq", "\\geq", "\\neq", "\\approx", "\\pm", "<", ">", "\\to", "\\in"]
_F_BIG = ["\\sum", "\\int", "\\prod"]
_F_FUNCS = ["\\sin", "\\cos", "\\log", "\\exp", "\\max", "\\min", "\\lim"]


def random_formula(rng: np.random.Generator, max_tokens: int = 14) -> str:
    """A LaTeX formula (without the dollar signs) from a small grammar that matplotlib's mathtext renders: variables,
    Greek letters, numbers, sub / superscripts, fractions, roots, big operators with limits, functions, relations.
    Tokens are separated by single spaces, braces are minimal: the string is the recogniser's training target."""

    def atom(depth: int) -> str:
        r = rng.random()
        if r < 0.40:
            a = _F_VARS[int(rng.integers(len(_F_VARS)))]
        elif r < 0.58:
            a = _F_GREEK[int(rng.integers(len(_F_GREEK)))]
        elif r < 0.72:
            a = str(int(rng.integers(0, 100)))
        elif r < 0.80 and depth < 2:
            a = "\\frac{%s}{%s}" % (expr(depth + 1, 2), expr(depth + 1, 2))
        elif r < 0.86 and depth < 2:
            a = "\\sqrt{%s}" % expr(depth + 1, 2)
        elif r < 0.92:
            a = "%s(%s)" % (_F_FUNCS[int(rng.integers(len(_F_FUNCS)))], _F_VARS[int(rng.integers(len(_F_VARS)))])
        elif depth < 2:
            a = "(%s)" % expr(depth + 1, 3)
        else:
            a = _F_VARS[int(rng.integers(len(_F_VARS)))]
        if a[0] not in "(\\" or a.startswith(("\\alpha", "\\beta", "\\gamma", "\\lambda", "\\sigma", "\\mu")):
            q = rng.random()
            if q < 0.22:
                a += "_{%s}" % (_F_VARS[int(rng.integers(len(_F_VARS)))] if rng.random() < 0.6 else str(int(rng.integers(0, 10))))
            if 0.15 < q < 0.35:
                a += "^{%s}" % (str(int(rng.integers(2, 5))) if rng.random() < 0.7 else _F_VARS[int(rng.integers(len(_F_VARS)))])
        return a

    def expr(depth: int, n: int) -> str:
        parts = [atom(depth)]
        for _ in range(int(rng.integers(0, n))):
            parts.append(_F_OPS[int(rng.integers(0, 5 if depth else len(_F_OPS)))])
            parts.append(atom(depth))
        return " ".join(parts)

    n = int(rng.integers(2, max(3, max_tokens // 3)))
    body = expr(0, n)
    if rng.random() < 0.3:
        big = _F_BIG[int(rng.integers(len(_F_BIG)))]
        v = "ijkn"[int(rng.integers(4))]
        body = "%s_{%s=%d}^{%s} %s" % (big, v, int(rng.integers(0, 2)), "nNmK"[int(rng.integers(4))], body)
    if "=" not in body and rng.random() < 0.7:
        body = "%s = %s" % (atom(1), body)
    return body


def random_inline_formula(rng: np.random.Generator) -> str:
    """A short formula of the kind that sits inside a sentence: one to three atoms joined by operators, sub / superscripts,
    Greek letters, function calls - nothing taller than a text line (no fractions, no big operators)."""

    def atom() -> str:
        r = rng.random()
        if r < 0.45:
            a = _F_VARS[int(rng.integers(len(_F_VARS)))]
        elif r < 0.70:
            a = _F_GREEK[int(rng.integers(len(_F_GREEK)))]
        elif r < 0.82:
            a = str(int(rng.integers(0, 100)))
        elif r < 0.92:
            a = "%s(%s)" % (_F_FUNCS[int(rng.integers(len(_F_FUNCS)))], _F_VARS[int(rng.integers(len(_F_VARS)))])
        else:
            a = "\\sqrt{%s}" % _F_VARS[int(rng.integers(len(_F_VARS)))]
        if a[0] not in "(\\0123456789" or a.startswith(("\\alpha", "\\beta", "\\gamma", "\\lambda", "\\sigma", "\\mu")):
            q = rng.random()
            if q < 0.35:
                a += "_{%s}" % (_F_VARS[int(rng.integers(len(_F_VARS)))] if rng.random() < 0.6 else str(int(rng.integers(0, 10))))
            if 0.25 < q < 0.5:
                a += "^{%s}" % (str(int(rng.integers(2, 5))) if rng.random() < 0.7 else _F_VARS[int(rng.integers(len(_F_VARS)))])
        return a

    n = int(rng.integers(1, 4))
    parts = [atom()]
    for _ in range(n - 1):
        parts.append(_F_OPS[int(rng.integers(len(_F_OPS)))])
        parts.append(atom())
    body = " ".join(parts)
    if n == 1 and len(body) <= 2:                      # a lone letter or number is not recognisable as mathematics
        body = "%s %s %s" % (body, _F_OPS[int(rng.integers(len(_F_OPS)))], atom())
    return body


_MATH_PARSER = None
_MATH_LOCK = None


def render_formula(latex: str, px: int = 42, fontset: str = "cm") -> Optional[np.ndarray]:
    """LaTeX -> uint8 [h, w] coverage (0 = paper, 255 = full ink), tightly cropped, through matplotlib's mathtext.
    ``px`` is the height of an ordinary letter's em box in pixels.  None when mathtext cannot parse the string."""
    r = render_formula_baseline(latex, px, fontset)
    return None if r is None else r[0]


def render_formula_baseline(latex: str, px: int = 42, fontset: str = "cm") -> Optional[Tuple[np.ndarray, int]]:
    """As ``render_formula`` plus the row of the crop on which the text baseline lies (for setting a formula into a line)."""
    global _MATH_PARSER, _MATH_LOCK
    import threading

    from matplotlib.font_manager import FontProperties
    from matplotlib.mathtext import MathTextParser

    if _MATH_LOCK is None:
        _MATH_LOCK = threading.Lock()
    with _MATH_LOCK:                       # the parser caches glyphs in shared state
        if _MATH_PARSER is None:
            _MATH_PARSER = MathTextParser("agg")
        try:
            r = _MATH_PARSER.parse("$" + latex + "$", dpi=72, prop=FontProperties(size=px, math_fontfamily=fontset))
        except Exception:  # noqa: BLE001 - pyparsing errors of many kinds
            return None
        a = np.asarray(r.image, dtype=np.uint8).copy()
        depth = float(r.depth)
    ys, xs = np.nonzero(a > 8)
    if ys.size == 0:
        return None
    return a[ys.min() : ys.max() + 1, xs.min() : xs.max() + 1], int(round(a.shape[0] - depth)) - int(ys.min())


LAYOUTS = ("single", "two_col", "two_col_title", "figure")
FORMULA_FONTSETS = ("cm", "stix", "dejavusans", "dejavuserif", "stixsans")


def render_layout_page(seed: int, size: Tuple[int, int] = A4_300DPI, layout: Optional[str] = None,
                       fonts: Optional[Sequence[str]] = None, font_px: Optional[int] = None, skew: float = 0.0,
                       noise: float = 2.0, blur: float = 0.0, margin: Optional[int] = None,
                       header_footer: Optional[bool] = None, formulas: float = 0.0,
                       i18n: float = 0.0, script: Optional[str] = None, inline: float = 0.0) -> Tuple[np.ndarray, PageTruth]:
    """A synthetic scanned page with a real layout: running header / footer, an optional full-width title, one or
    two columns of paragraphs, optionally a figure with its caption inside a column; one font per page drawn from
    ``fonts``; optional skew (degrees), blur and sensor noise.  ``formulas`` > 0: that share of the paragraph breaks
    carries a centred display formula (``random_formula`` through mathtext).  ``i18n``: probability that the page is
    written in the Western European vocabulary (accented letters, ß, œ, € ...) mixed with the English one.
    ``script``: ``"cyrillic"`` / ``"greek"`` writes the page in the Russian / Greek vocabulary of ``utils.scripts`` (one
    word in ten stays English, as in real documents) with a font that has the alphabet.  ``inline``: that share of the
    paragraph lines carries one or two short formulas inside the sentence (``random_inline_formula``, set on the text
    baseline; the line's truth text holds them as ``$...$``).  The truth lists the lines in READING
    order (header, title, first column top to bottom, second column, footer) in the upright frame plus ``angle``."""
    W, H = size
    rng = np.random.default_rng(seed)
    layout = layout or LAYOUTS[int(rng.integers(len(LAYOUTS)))]
    fonts = available_fonts(fonts or TRAIN_FONTS) or ["DejaVuSans.ttf"]
    if script in ("cyrillic", "greek"):
        probe = ord("Ж") if script == "cyrillic" else ord("λ")
        fonts = [n for n in fonts if not n.startswith("cm") and probe in (font_code_points(n) or (probe,))] or ["DejaVuSans.ttf"]
    fname = fonts[int(rng.integers(len(fonts)))]
    scale = W / 2480.0
    px = font_px or int(rng.integers(30, 50) * scale)
    px = max(px, 8)
    margin = int((margin if margin is not None else 200) * scale) if margin is None or margin > 0 else 0
    margin = max(margin, 6)
    paper = int(rng.integers(228, 252))
    img = Image.new("L", (W, H), paper)
    d = ImageDraw.Draw(img)
    f = named_font(fname, px)
    # headings: the bold sibling where one exists, else the same face larger
    bold_name = {"DejaVuSans.ttf": "DejaVuSans-Bold.ttf", "DejaVuSerif.ttf": "DejaVuSerif-Bold.ttf",
                 "DejaVuSansMono.ttf": "DejaVuSansMono-Bold.ttf", "STIXGeneral.ttf": "STIXGeneralBol.ttf",
                 "Lato-Regular.ttf": "Lato-Bold.ttf", "cmr10.ttf": "cmb10.ttf", "Lato-Light.ttf": "Lato-Semibold.ttf",
                 "Lato-Medium.ttf": "Lato-Heavy.ttf", "Lato-LightItalic.ttf": "Lato-SemiboldItalic.ttf",
                 "Lato-MediumItalic.ttf": "Lato-HeavyItalic.ttf", "STIXGeneralItalic.ttf": "STIXGeneralBolIta.ttf",
                 "DejaVuSans-Oblique.ttf": "DejaVuSans-BoldOblique.ttf",
                 "DejaVuSerif-Italic.ttf": "DejaVuSerif-BoldItalic.ttf"}.get(fname, fname)
    fb = named_font(bold_name if font_path(bold_name) else fname, int(px * 1.3))
    line_h = int(px * rng.uniform(1.45, 1.7))
    ink = lambda: int(rng.integers(5, 80))
    boxes, texts, blocks, figures = [], [], [], []
    fboxes, ftexts, fpos = [], [], []
    fontset = FORMULA_FONTSETS[int(rng.integers(len(FORMULA_FONTSETS)))]
    block_id = [0]
    vocab = None
    if i18n > 0 and rng.random() < i18n:
        vocab = list(_WORDS_I18N) * 2 + list(_WORDS)          # two thirds Western European words
    if script in ("cyrillic", "greek"):
        from ..utils import scripts as _scripts

        own = list(_scripts.WORDS_RU if script == "cyrillic" else _scripts.WORDS_EL)
        vocab = own * (1 + (9 * len(_WORDS)) // len(own)) + list(_WORDS)
    random_text = lambda r, *a, **k: globals()["random_text"](r, *a, vocabulary=vocab, **k)    # noqa: E731

    def put(x: int, y: int, t: str, ff, max_x: int) -> Optional[int]:
        t = _font_safe(t, fname)
        if not t:
            return None
        bb = d.textbbox((x, y), t, font=ff)
        while bb[2] > max_x and " " in t:        # trim words until the line fits its column
            t = t.rsplit(" ", 1)[0]
            bb = d.textbbox((x, y), t, font=ff)
        if bb[2] > max_x or not t:
            return None
        d.text((x, y), t, fill=ink(), font=ff)
        boxes.append([bb[0] - 3, bb[1] - 3, bb[2] + 3, bb[3] + 3])
        texts.append(t)
        blocks.append(block_id[0])
        return bb[3]

    iboxes, itexts, iline, line_parts = [], [], [], {}
    ascent = f.getmetrics()[0]
    space = max(2, int(round(f.getlength("a b") - f.getlength("ab"))))

    def put_inline(x: int, y: int, max_x: int, max_chars: int) -> bool:
        """words, formula, words (, formula, words) on one baseline; False when the pieces do not fit."""
        nf = 1 if rng.random() < 0.7 else 2
        maths = []
        for _ in range(nf):
            latex = random_inline_formula(rng)
            r = render_formula_baseline(latex, int(round(px * rng.uniform(0.95, 1.1))), fontset)
            if r is None or r[0].shape[1] < px or r[0].shape[1] > 0.4 * (max_x - x):
                return False
            maths.append((latex, r[0], r[1]))
        free = (max_x - x) - sum(m[1].shape[1] + 2 * space for m in maths)
        budget = int(free / avg_char) - 2
        if budget < 6 * (nf + 1):
            return False
        cuts = sorted(int(v) for v in rng.uniform(0.12, 0.88, size=nf) * budget)
        shares = [cuts[0]] + [b_ - a_ for a_, b_ in zip(cuts, cuts[1:])] + [budget - cuts[-1]]
        col = ink()
        parts, cur = [], x
        y_lo, y_hi = 1 << 30, -1
        for k in range(nf + 1):
            t = _font_safe(random_text(rng, max(3, shares[k]), min_words=1), fname).replace("$", "")
            if k == 0 and rng.random() < 0.12:
                t = ""                                      # the line begins with the formula
            if k == nf and rng.random() < 0.12:
                t = ""                                      # ... or ends with it
            if t:
                bb = d.textbbox((cur, y), t, font=f)
                if bb[2] > max_x:
                    return False
                parts.append(("t", bb[0], bb[2], t, bb[1], bb[3]))
                cur = bb[2] + space
            if k < nf:
                latex, cov, base = maths[k]
                top = y + ascent - base
                parts.append(("m", cur, cur + cov.shape[1], latex, top, top + cov.shape[0], cov))
                cur += cov.shape[1] + space
        if cur - space > max_x or not any(p_[0] == "t" for p_ in parts):
            return False
        for p_ in parts:
            if p_[0] == "t":
                d.text((p_[1], y), p_[3], fill=col, font=f)
            else:
                cov = p_[6].astype(np.float32) / 255.0
                reg = np.asarray(img.crop((p_[1], p_[4], p_[2], p_[5])), dtype=np.float32)
                img.paste(Image.fromarray((reg * (1 - cov) + col * cov).astype(np.uint8)), (p_[1], p_[4]))
            y_lo, y_hi = min(y_lo, p_[4]), max(y_hi, p_[5])
        ty = [p_ for p_ in parts if p_[0] == "t"]
        t_lo, t_hi = min(p_[4] for p_ in ty), max(p_[5] for p_ in ty)
        li = len(texts)
        boxes.append([parts[0][1] - 3, y_lo - 3, parts[-1][2] + 3, y_hi + 3])
        texts.append(" ".join(p_[3] if p_[0] == "t" else "$" + p_[3] + "$" for p_ in parts))
        blocks.append(block_id[0])
        line_parts[li] = [(p_[0], float(p_[1]), float(p_[2]), p_[3]) for p_ in parts]
        for p_ in parts:
            if p_[0] == "m":
                iboxes.append([p_[1] - 2, min(t_lo, p_[4]) - 2, p_[2] + 2, max(t_hi, p_[5]) + 2])
                itexts.append(p_[3])
                iline.append(li)
        return True

    avg_char = max(1.0, f.getlength("abcdefghij") / 10.0)
    with_hf = bool(rng.random() < 0.6) if header_footer is None else header_footer
    top, bottom = margin, H - margin
    if with_hf:
        put(margin, int(0.025 * H), random_text(rng, 40).title(), named_font(fname, max(8, int(px * 0.8))), W - margin)
        block_id[0] += 1
        top = max(top, int(0.095 * H))
        bottom = min(bottom, int(0.905 * H))
    y = top
    if layout == "two_col_title" or (layout != "single" and rng.random() < 0.3):
        yb = put(margin, y, random_text(rng, max(8, int((W - 2 * margin) / avg_char / 1.6))).title(), fb, W - margin)
        block_id[0] += 1
        y = (yb if yb is not None else y) + int(1.6 * line_h)
    ncol = 1 if layout in ("single",) else 2
    if layout == "figure":
        ncol = int(rng.integers(1, 3))
    gutter = int(rng.uniform(3.0, 5.0) * px)
    col_w = (W - 2 * margin - (ncol - 1) * gutter) // ncol
    fig_col = int(rng.integers(ncol)) if layout == "figure" else -1
    n_fig = 0
    for c in range(ncol):
        x0 = margin + c * (col_w + gutter)
        x1 = x0 + col_w
        max_chars = max(6, int(col_w / avg_char))
        yc = y
        fig_at = yc + int(rng.uniform(0.1, 0.5) * (bottom - yc)) if c == fig_col else None
        first = True
        while yc + line_h < bottom:
            if fig_at is not None and yc >= fig_at:
                fh = int(rng.uniform(0.12, 0.25) * H)
                if yc + fh + 2 * line_h < bottom:
                    fw = int(rng.uniform(0.7, 1.0) * col_w)
                    fx = x0 + (col_w - fw) // 2
                    yc += line_h // 2
                    img.paste(Image.fromarray(_figure_patch(rng, fw, fh)), (fx, yc))
                    if rng.random() < 0.5:
                        d.rectangle([fx, yc, fx + fw - 1, yc + fh - 1], outline=ink(), width=max(1, px // 16))
                    figures.append([fx, yc, fx + fw, yc + fh])
                    n_fig += 1
                    yc += fh + line_h // 2
                    block_id[0] += 1
                    yb = put(x0, yc, f"Figure {n_fig}: " + random_text(rng, max(6, max_chars - 12)), f, x1)
                    block_id[0] += 1
                    yc = (yb if yb is not None else yc) + line_h
                fig_at = None
                continue
            if first or rng.random() < 0.06:
                t = random_text(rng, max(6, max_chars // 2)).title()
                first = False
                block_id[0] += 1
                yc += line_h // 3
                yb = put(x0, yc, t, fb if ncol == 1 else named_font(bold_name if font_path(bold_name) else fname, int(px * 1.1)), x1)
                yc = (yb if yb is not None else yc) + int(0.9 * line_h)
                block_id[0] += 1
                continue
            frac = 1.0 if rng.random() > 0.16 else float(rng.uniform(0.3, 0.8))
            if not (inline > 0 and frac == 1.0 and rng.random() < inline and put_inline(x0, yc, x1, max_chars)):
                t = random_text(rng, max(6, int(max_chars * frac)), min_words=2)
                put(x0 + (0 if rng.random() > 0.1 else int(rng.integers(1, 4) * px)), yc, t, f, x1)
            yc += line_h
            if frac < 1.0 or rng.random() < 0.05:
                yc += int(rng.uniform(0.9, 1.4) * line_h)       # paragraph gap
                block_id[0] += 1
                if formulas > 0 and rng.random() < formulas:
                    latex = random_formula(rng)
                    fm = render_formula(latex, int(px * rng.uniform(1.0, 1.25)), fontset)
                    if fm is not None and 3 * px <= fm.shape[1] < col_w - 2 * px and yc + fm.shape[0] + 2 * line_h < bottom:
                        fx = x0 + (col_w - fm.shape[1]) // 2
                        cov = fm.astype(np.float32) / 255.0
                        reg = np.asarray(img.crop((fx, yc, fx + fm.shape[1], yc + fm.shape[0])), dtype=np.float32)
                        img.paste(Image.fromarray((reg * (1 - cov) + ink() * cov).astype(np.uint8)), (fx, yc))
                        fboxes.append([fx - 4, yc - 4, fx + fm.shape[1] + 4, yc + fm.shape[0] + 4])
                        ftexts.append(latex)
                        fpos.append(len(texts))
                        yc += fm.shape[0] + int(rng.uniform(0.9, 1.4) * line_h)
                        block_id[0] += 1
        block_id[0] += 1
    if with_hf:
        t = f"{int(rng.integers(1, 400))}" if rng.random() < 0.5 else "Page %d of %d" % (int(rng.integers(1, 50)), 50)
        put(W // 2 - 2 * px, int(0.945 * H), t, named_font(fname, max(8, int(px * 0.8))), W - margin)
    if blur > 0:
        from PIL import ImageFilter

        img = img.filter(ImageFilter.GaussianBlur(blur))
    if abs(skew) > 1e-3:
        img = img.rotate(skew, resample=Image.BICUBIC, fillcolor=paper)
    arr = np.asarray(img, dtype=np.uint8)
    if noise > 0:
        nz = rng.normal(0.0, noise, size=(H // 4 + 1, W // 4 + 1)).astype(np.float32)
        nz = np.repeat(np.repeat(nz, 4, axis=0), 4, axis=1)[:H, :W]
        arr = np.clip(arr.astype(np.float32) + nz, 0, 255).astype(np.uint8)
    truth = PageTruth(np.asarray(boxes, dtype=np.float32).reshape(-1, 4), texts,
                      np.asarray(figures, dtype=np.float32).reshape(-1, 4), blocks, float(skew), layout, [fname],
                      np.asarray(fboxes, dtype=np.float32).reshape(-1, 4), ftexts, fpos,
                      np.asarray(iboxes, dtype=np.float32).reshape(-1, 4), itexts, iline, line_parts)
    return arr, truth


def render_table_page(seed: int, size: Tuple[int, int] = A4_300DPI, rows: int = 5, cols: int = 3, ruled: bool = True,
                      row_pitch: float = 1.6, column: bool = False, font_name: str = "DejaVuSans.ttf",
                      font_px: int = 38, wrap: float = 0.0) -> Tuple[np.ndarray, PageTruth, List[List[str]]]:
    """A page with two paragraphs and a ``rows`` x ``cols`` table between them (first column labels, the others
    numbers or short words; optional ruling lines; ``row_pitch`` line heights from row to row).  ``column``: the page
    has two text columns and the table sits inside the left one.  ``wrap``: that share of the body rows has a label
    that runs onto a second line (at line spacing under the first; the row is a line taller).  The truth lists the
    lines in reading order with the cells row by row (the second line of a label after its row); the third value is the
    grid of cell texts."""
    W, H = size
    rng = np.random.default_rng(seed)
    scale = W / 2480.0
    px = max(8, int(font_px * scale))
    margin = max(6, int(200 * scale))
    img = Image.new("L", (W, H), 244)
    d = ImageDraw.Draw(img)
    f = named_font(font_name, px)
    line_h = int(px * 1.55)
    avg_char = max(1.0, f.getlength("abcdefghij") / 10.0)
    boxes, texts, blocks = [], [], []
    blk = [0]

    def put(x, y, t):
        bb = d.textbbox((x, y), t, font=f)
        d.text((x, y), t, fill=30, font=f)
        boxes.append([bb[0] - 3, bb[1] - 3, bb[2] + 3, bb[3] + 3])
        texts.append(t)
        blocks.append(blk[0])

    gutter = 4 * px
    col_w = (W - 2 * margin - gutter) // 2 if column else W - 2 * margin
    max_chars = max(8, int(col_w / avg_char) - 2)

    def paragraph(x, y, n):
        for _ in range(n):
            put(x, y, random_text(rng, max_chars, min_words=2))
            y += line_h
        blk[0] += 1
        return y

    y = paragraph(margin, margin, 4) + line_h
    grid: List[List[str]] = []
    cw = col_w // cols
    y_top = y
    for r in range(rows):
        row = []
        for c in range(cols):
            if r == 0:
                t = random_text(rng, 10).title() or "Item"
            elif c == 0:
                t = random_text(rng, 12) or "row"
            else:
                t = ("%d.%02d" % (int(rng.integers(1, 5000)), int(rng.integers(100)))) if rng.random() < 0.7 else (random_text(rng, 8) or "n/a")
            t = t[: max(3, int((cw - 2.5 * px) / avg_char))].strip() or "x"     # a typeset table keeps a gutter between cells
            put(margin + c * cw + px // 2, y, t)
            row.append(t)
        grid.append(row)
        if wrap > 0 and r > 0 and rng.random() < wrap:
            t2 = (random_text(rng, 12) or "more")[: max(3, int((cw - 2.5 * px) / avg_char))].strip() or "more"
            y += line_h
            put(margin + px // 2, y, t2)
            row[0] = row[0] + " " + t2
        if ruled:
            d.line([margin, y + int(line_h * row_pitch) - px // 3, margin + cols * cw, y + int(line_h * row_pitch) - px // 3], fill=90, width=max(1, px // 16))
        y += int(line_h * row_pitch)
    if ruled:
        for c in range(cols + 1):
            d.line([margin + c * cw, y_top - px // 3, margin + c * cw, y - px // 3], fill=90, width=max(1, px // 16))
        d.line([margin, y_top - px // 3, margin + cols * cw, y_top - px // 3], fill=90, width=max(1, px // 16))
    blk[0] += 1
    y = paragraph(margin, y + line_h, 4)
    if column:
        paragraph(margin + col_w + gutter, margin, 14)
    arr = np.asarray(img, dtype=np.uint8)
    nz = rng.normal(0.0, 2.0, size=(H // 4 + 1, W // 4 + 1)).astype(np.float32)
    arr = np.clip(arr.astype(np.float32) + np.repeat(np.repeat(nz, 4, axis=0), 4, axis=1)[:H, :W], 0, 255).astype(np.uint8)
    return arr, PageTruth(np.asarray(boxes, dtype=np.float32).reshape(-1, 4), texts, blocks=blocks, layout="table",
                          font_names=[font_name]), grid


def render_line(text: str, height: int = 48, font_px: int = 34, seed: int = 0, pad: int = 6,
                bold: bool = False) -> np.ndarray:
    """One text line as uint8 [height, w] (dark ink on light paper), tightly cropped."""
    rng = np.random.default_rng(seed)
    f = font(font_px, bold)
    w = int(f.getlength(text)) + 2 * pad + 2
    img = Image.new("L", (max(w, 8), height), int(rng.integers(225, 256)))
    d = ImageDraw.Draw(img)
    bb = d.textbbox((0, 0), text, font=f)
    y = (height - (bb[3] - bb[1])) // 2 - bb[1]
    d.text((pad, y), text, fill=int(rng.integers(0, 70)), font=f)
    return np.asarray(img, dtype=np.uint8)


# --------------------------------------------------------------------------------------- PDF writing
class _PdfWriter:
    """Minimal PDF 1.4 writer (classic xref table) or 1.5 (object streams + xref stream)."""

    def __init__(self) -> None:
        self.objs: List[bytes] = []  # object bodies, 1-based ids

    def add(self, body: bytes) -> int:
        self.objs.append(body)
        return len(self.objs)

    def reserve(self) -> int:
        self.objs.append(b"null")
        return len(self.objs)

    def set(self, oid: int, body: bytes) -> None:
        self.objs[oid - 1] = body

    @staticmethod
    def stream(dict_entries: bytes, data: bytes) -> bytes:
        return b"<< " + dict_entries + b" /Length " + str(len(data)).encode() + b" >>\nstream\n" + data + b"\nendstream"

    def finish(self, root: int, info: Optional[int] = None, xref_stream: bool = False) -> bytes:
        out = io.BytesIO()
        out.write(b"%PDF-1.5\n%\xe2\xe3\xcf\xd3\n" if xref_stream else b"%PDF-1.4\n%\xe2\xe3\xcf\xd3\n")
        offsets = [0] * (len(self.objs) + 1)
        for i, body in enumerate(self.objs, start=1):
            offsets[i] = out.tell()
            out.write(f"{i} 0 obj\n".encode() + body + b"\nendobj\n")
        n = len(self.objs) + 1
        if not xref_stream:
            xref_pos = out.tell()
            out.write(f"xref\n0 {n}\n".encode())
            out.write(b"0000000000 65535 f \n")
            for i in range(1, n):
                out.write(f"{offsets[i]:010d} 00000 n \n".encode())
            tr = f"trailer\n<< /Size {n} /Root {root} 0 R" + (f" /Info {info} 0 R" if info else "") + " >>\n"
            out.write(tr.encode())
            out.write(f"startxref\n{xref_pos}\n%%EOF\n".encode())
        else:
            xid = n
            xref_pos = out.tell()
            rows = bytearray()
            rows += bytes([0, 0, 0, 0, 0, 255])  # object 0: free
            for i in range(1, n):
                rows += bytes([1]) + offsets[i].to_bytes(4, "big") + bytes([0])
            rows += bytes([1]) + xref_pos.to_bytes(4, "big") + bytes([0])
            data = zlib.compress(bytes(rows))
            d = (f"/Type /XRef /Size {n + 1} /W [1 4 1] /Root {root} 0 R".encode()
                 + (f" /Info {info} 0 R".encode() if info else b"") + b" /Filter /FlateDecode")
            out.write(f"{xid} 0 obj\n".encode() + self.stream(d, data) + b"\nendobj\n")
            out.write(f"startxref\n{xref_pos}\n%%EOF\n".encode())
        return out.getvalue()


def encode_page_jpeg(page: np.ndarray, quality: int = 85, restart_blocks: int = 62) -> bytes:
    """Baseline JPEG with a restart marker every ``restart_blocks`` MCUs (0: none).

    Restart intervals make the entropy-coded segment decodable in parallel, which is
    what the GPU decoder (csrc/kernels/jpeg_decode.hip) uses; files without them are
    decoded on the host thread pool instead."""
    im = Image.fromarray(page)
    buf = io.BytesIO()
    kw = dict(format="JPEG", quality=quality, optimize=False, progressive=False)
    if restart_blocks > 0:
        kw["restart_marker_blocks"] = restart_blocks
    im.save(buf, **kw)
    return buf.getvalue()


def encode_page_ccitt_g4(page: np.ndarray, threshold: int = 128) -> Tuple[bytes, bool]:
    """Gray page -> (Group 4 fax strip, BlackIs1 flag for the PDF image) through PIL's libtiff writer (one strip)."""
    import io

    from PIL import Image

    gray = page if page.ndim == 2 else page.mean(axis=2)
    im = Image.fromarray(((gray >= threshold) * 255).astype(np.uint8)).convert("1")
    buf = io.BytesIO()
    im.save(buf, format="TIFF", compression="group4", tiffinfo={278: im.height})
    raw = buf.getvalue()
    t = Image.open(io.BytesIO(raw))
    offs, cnts = t.tag_v2[273], t.tag_v2[279]
    assert len(offs) == 1, "expected a single strip"
    # photometric 1 (BlackIsZero): the coder's "white" runs are the file's 0 bits = black pixels
    return raw[offs[0] : offs[0] + cnts[0]], int(t.tag_v2.get(262, 0)) == 1


def png_predict_rows(page: np.ndarray, row_filter: Optional[int] = None) -> bytes:
    """Rows of ``page`` (uint8 [H, W] or [H, W, 3]) with PNG predictors applied, each prefixed by its filter-type
    byte (PDF ``/Predictor 15``).  ``row_filter``: 0 None, 1 Sub, 2 Up, 3 Average, 4 Paeth; default: row y uses y % 5."""
    H = page.shape[0]
    bpp = 1 if page.ndim == 2 else page.shape[2]
    flat = page.reshape(H, -1).astype(np.int32)
    out = bytearray()
    prev = np.zeros(flat.shape[1], np.int32)
    zero = np.zeros(bpp, np.int32)
    for y in range(H):
        cur = flat[y]
        a = np.concatenate([zero, cur[:-bpp]])
        ft = y % 5 if row_filter is None else row_filter
        if ft == 0:
            enc = cur
        elif ft == 1:
            enc = cur - a
        elif ft == 2:
            enc = cur - prev
        elif ft == 3:
            enc = cur - ((a + prev) >> 1)
        else:
            c = np.concatenate([zero, prev[:-bpp]])
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            enc = cur - np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        out.append(ft)
        out += (enc & 255).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def make_scanned_pdf(pages: Sequence[np.ndarray], codec: str = "jpeg", quality: int = 85, dpi: int = 300,
                     restart_blocks: int = 62, xref_stream: bool = False,
                     encoded: Optional[Sequence[bytes]] = None, rotate: int = 0, predictor: int = 1,
                     encoded_ccitt: Optional[Sequence[Tuple[bytes, bool]]] = None) -> bytes:
    """PDF with one full-page image per page.  ``pages``: uint8 [H, W] (gray) or [H, W, 3] arrays.
    ``predictor`` (codec "flate"): 1 none; 10-15: rows are PNG-predicted before compression, as image tools write them.

    ``encoded`` may carry pre-encoded image streams (same order) to avoid re-encoding.
    ``rotate`` (0/90/180/270): what a scanner's "landscape feed" produces - the stored raster is turned
    counter-clockwise by that angle and the page carries ``/Rotate`` so that a viewer shows it upright."""
    assert rotate in (0, 90, 180, 270)
    if rotate:
        assert encoded is None
        pages = [np.ascontiguousarray(np.rot90(pg, k=rotate // 90)) for pg in pages]
    w = _PdfWriter()
    catalog = w.reserve()
    pages_id = w.reserve()
    kids = []
    for i, pg in enumerate(pages):
        H, W = pg.shape[:2]
        cs = b"/DeviceGray" if pg.ndim == 2 else b"/DeviceRGB"
        if codec == "jpeg":
            data = encoded[i] if encoded is not None else encode_page_jpeg(pg, quality, restart_blocks)
            flt = b"/DCTDecode"
        elif codec == "flate":
            if encoded is not None:
                data = encoded[i]
            elif predictor >= 10:
                data = zlib.compress(png_predict_rows(pg, None if predictor == 15 else predictor - 10), 1)
            else:
                data = zlib.compress(np.ascontiguousarray(pg).tobytes(), 1)
            flt = b"/FlateDecode"
            if predictor >= 10:
                flt += b" /DecodeParms << /Predictor %d /Colors %d /BitsPerComponent 8 /Columns %d >>" % (
                    predictor, 1 if pg.ndim == 2 else 3, W)
        elif codec == "ccitt":
            # bilevel Group 4 fax data, what office scanners emit for black-and-white scans
            data, black_is_1 = encoded_ccitt[i] if encoded_ccitt is not None else encode_page_ccitt_g4(pg)
            img = w.add(w.stream(
                b"/Type /XObject /Subtype /Image /Width %d /Height %d /ColorSpace /DeviceGray /BitsPerComponent 1 "
                b"/Filter /CCITTFaxDecode /DecodeParms << /K -1 /Columns %d /Rows %d /BlackIs1 %s >>"
                % (W, H, W, H, b"true" if black_is_1 else b"false"), data))
            flt = None
        else:
            raise ValueError("codec must be jpeg, flate or ccitt")
        if flt is not None:
            img = w.add(w.stream(
                b"/Type /XObject /Subtype /Image /Width %d /Height %d /ColorSpace %s /BitsPerComponent 8 /Filter %s"
                % (W, H, cs, flt), data))
        wpt, hpt = W * 72.0 / dpi, H * 72.0 / dpi
        content = b"q %.2f 0 0 %.2f 0 0 cm /Im0 Do Q" % (wpt, hpt)
        cid = w.add(w.stream(b"", content))
        pid = w.add(
            b"<< /Type /Page /Parent %d 0 R /MediaBox [0 0 %.2f %.2f] /Resources << /XObject << /Im0 %d 0 R >> >> "
            b"/Contents %d 0 R%s >>" % (pages_id, wpt, hpt, img, cid, b" /Rotate %d" % rotate if rotate else b""))
        kids.append(pid)
    w.set(pages_id, b"<< /Type /Pages /Count %d /Kids [%s] >>" % (len(kids), b" ".join(b"%d 0 R" % k for k in kids)))
    w.set(catalog, b"<< /Type /Catalog /Pages %d 0 R >>" % pages_id)
    return w.finish(catalog, xref_stream=xref_stream)


def _pdf_escape(s: str) -> bytes:
    b = s.encode("latin-1", "replace")
    return b.replace(b"\\", b"\\\\").replace(b"(", b"\\(").replace(b")", b"\\)")


def make_text_pdf(pages: Sequence[Sequence[str]], font_size: float = 11.0, compress: bool = True,
                  xref_stream: bool = False, use_tj_arrays: bool = False) -> bytes:
    """Born-digital PDF: each page is a list of text lines set in Helvetica (WinAnsi)."""
    w = _PdfWriter()
    catalog = w.reserve()
    pages_id = w.reserve()
    font_id = w.add(b"<< /Type /Font /Subtype /Type1 /BaseFont /Helvetica /Encoding /WinAnsiEncoding >>")
    kids = []
    for lines in pages:
        ops = [b"BT", b"/F1 %.1f Tf" % font_size, b"%.1f TL" % (font_size * 1.4), b"72 770 Td"]
        for ln in lines:
            if use_tj_arrays and " " in ln:
                parts = ln.split(" ")
                arr = b" -278 ".join(b"(" + _pdf_escape(p) + b")" for p in parts)
                ops.append(b"[" + arr + b"] TJ T*")
            else:
                ops.append(b"(" + _pdf_escape(ln) + b") Tj T*")
        ops.append(b"ET")
        content = b"\n".join(ops)
        if compress:
            cid = w.add(w.stream(b"/Filter /FlateDecode", zlib.compress(content)))
        else:
            cid = w.add(w.stream(b"", content))
        pid = w.add(
            b"<< /Type /Page /Parent %d 0 R /MediaBox [0 0 595.28 841.89] "
            b"/Resources << /Font << /F1 %d 0 R >> >> /Contents %d 0 R >>" % (pages_id, font_id, cid))
        kids.append(pid)
    w.set(pages_id, b"<< /Type /Pages /Count %d /Kids [%s] >>" % (len(kids), b" ".join(b"%d 0 R" % k for k in kids)))
    w.set(catalog, b"<< /Type /Catalog /Pages %d 0 R >>" % pages_id)
    return w.finish(catalog, xref_stream=xref_stream)


def make_text_pdf_columns(columns: Sequence[Sequence[str]], title: Optional[str] = None, font_size: float = 10.0,
                          header: Optional[str] = None, footer: Optional[str] = None, compress: bool = True) -> bytes:
    """Born-digital single page with a full-width title and N text columns.  The content stream interleaves the
    columns LINE BY LINE (row 0 of every column, then row 1, ...), the order in which a layout engine that fills a
    page top to bottom may emit them - only position-based analysis recovers the reading order."""
    w = _PdfWriter()
    catalog = w.reserve()
    pages_id = w.reserve()
    font_id = w.add(b"<< /Type /Font /Subtype /Type1 /BaseFont /Helvetica /Encoding /WinAnsiEncoding >>")
    W, H, margin = 595.28, 841.89, 56.0
    ncol = max(1, len(columns))
    gutter = 28.0
    col_w = (W - 2 * margin - (ncol - 1) * gutter) / ncol
    lead = font_size * 1.4
    ops = [b"BT", b"/F1 %.1f Tf" % font_size]
    if header:
        ops.append(b"1 0 0 1 %.2f %.2f Tm (%s) Tj" % (margin, H - 30.0, _pdf_escape(header)))
    y_top = H - 80.0
    if title:
        ops.append(b"/F1 %.1f Tf 1 0 0 1 %.2f %.2f Tm (%s) Tj /F1 %.1f Tf" % (font_size * 1.6, margin, y_top, _pdf_escape(title), font_size))
        y_top -= 3.0 * lead
    rows = max(len(c) for c in columns) if columns else 0
    for r in range(rows):
        for c, col in enumerate(columns):
            if r < len(col) and col[r]:
                ops.append(b"1 0 0 1 %.2f %.2f Tm (%s) Tj" % (margin + c * (col_w + gutter), y_top - r * lead, _pdf_escape(col[r])))
    if footer:
        ops.append(b"1 0 0 1 %.2f %.2f Tm (%s) Tj" % (W / 2 - 10.0, 28.0, _pdf_escape(footer)))
    ops.append(b"ET")
    content = b"\n".join(ops)
    cid = w.add(w.stream(b"/Filter /FlateDecode", zlib.compress(content)) if compress else w.stream(b"", content))
    pid = w.add(b"<< /Type /Page /Parent %d 0 R /MediaBox [0 0 %.2f %.2f] /Resources << /Font << /F1 %d 0 R >> >> "
                b"/Contents %d 0 R >>" % (pages_id, W, H, font_id, cid))
    w.set(pages_id, b"<< /Type /Pages /Count 1 /Kids [%d 0 R] >>" % pid)
    w.set(catalog, b"<< /Type /Catalog /Pages %d 0 R >>" % pages_id)
    return w.finish(catalog)
