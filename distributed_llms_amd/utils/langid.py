"""Language identification of extracted text.

The reference's engine (magic-pdf, services/ocr_service/ocr_service.py:215-220) looks at a document's language to
pick its OCR model; its model downloader fetches one multilingual recognizer
(services/ocr_service/download_models_hf.py:41).  Here the recogniser's character set is Latin, so the detected
language does not switch models; it is reported with every document (``DocumentResult.language``) so that a caller
can tell when a document is outside what the shipped weights read, and it picks the text-layer fallbacks.

Two stages, no model files:
* script: share of letters per Unicode block (Latin, Cyrillic, Greek, Arabic, Hebrew, Devanagari, Han, Kana,
  Hangul, Thai) - a non-Latin majority decides on its own (``ru``-like answers are by script: ``cyrillic``);
* Latin-script languages: function-word hit rate over a few dozen high-frequency words per language plus
  language-specific letters as tie-breakers.  Short texts (< 3 known words) are ``und`` (undetermined).
"""
from __future__ import annotations

import re
from collections import Counter
from typing import Dict, Tuple

import numpy as np

_SCRIPTS = (
    ("latin", ((0x41, 0x5A), (0x61, 0x7A), (0xC0, 0x24F))),
    ("cyrillic", ((0x400, 0x52F),)),
    ("greek", ((0x370, 0x3FF),)),
    ("arabic", ((0x600, 0x6FF), (0x750, 0x77F))),
    ("hebrew", ((0x590, 0x5FF),)),
    ("devanagari", ((0x900, 0x97F),)),
    ("thai", ((0xE00, 0xE7F),)),
    ("kana", ((0x3040, 0x30FF),)),
    ("hangul", ((0xAC00, 0xD7AF), (0x1100, 0x11FF))),
    ("han", ((0x4E00, 0x9FFF), (0x3400, 0x4DBF))),
)

_WORDS: Dict[str, str] = {
    "en": "the of and to in is that it was for on are with as be at this have from or by not but what all were when we "
          "there can an your which their said if will each about how up out them then she many some so these would other",
    "de": "der die und in den von zu das mit sich des auf für ist im dem nicht ein eine als auch es an werden aus er hat "
          "dass sie nach wird bei einer um am sind noch wie einem über einen so zum war haben nur oder aber vor zur bis",
    "fr": "de la le et les des en un du une que est pour qui dans a par plus pas au sur ne se ce il sont la avec son "
          "aux comme mais ou nous cette ses être fait ont leur aussi été elle vous très entre sans deux même",
    "es": "de la que el en y a los del se las por un para con no una su al lo como más pero sus le ya o este sí porque "
          "esta entre cuando muy sin sobre también me hasta hay donde quien desde todo nos durante todos uno",
    "it": "di e il la che in a per un è del le con non una si da i della sono al più come ma dei nel anche alla lo gli "
          "o se ha questo delle cui tra era fra loro essere quando molto suo tutti dopo due stato solo così",
    "pt": "de a o que e do da em um para é com não uma os no se na por mais as dos como mas foi ao ele das tem à seu "
          "sua ou ser quando muito há nos já está eu também só pelo pela até isso ela entre era depois sem mesmo",
    "nl": "de van het een en in is dat op te zijn met voor niet aan er om ook als dan maar bij of uit nog worden door "
          "naar heeft tot ze wordt over hij die deze zich kan meer wat was dit hebben werd geen na onder tegen",
    "sv": "och i att det som en på är av för med till den har de inte om ett han men var jag sig från vi så kan man "
          "när år säger hon under också efter eller nu sin där vid mot ska skulle kommer ut får finns vara hade alla",
    "pl": "w i na z do nie się że to jest o jak po a od przez za co dla ale tak czy jego są był tym oraz lub już tylko "
          "może być które ich przy pod też by bardzo było tego który jej roku ze przed między także będzie",
}
_LETTERS: Dict[str, str] = {"de": "äöüß", "fr": "éèêàçùœ", "es": "ñ¿¡óáí", "it": "èàòù", "pt": "ãõçâê", "sv": "åäö",
                            "pl": "ąęłńśźżć", "nl": "ĳ"}
_SETS = {k: frozenset(v.split()) for k, v in _WORDS.items()}
_TOKEN = re.compile(r"[^\W\d_]+", re.UNICODE)


_ALPHA = re.compile(r"[^\W\d_]+", re.UNICODE)


def _script_table() -> np.ndarray:
    """Script index (1-based, 0 = none) per BMP code point; the first script listing a code point keeps it."""
    lut = np.zeros(0x10000, dtype=np.uint8)
    for k, (_, ranges) in enumerate(_SCRIPTS):
        for lo, hi in ranges:
            seg = lut[lo : hi + 1]
            seg[seg == 0] = k + 1
    return lut


_SCRIPT_LUT = _script_table()


def _shares_of(cp: np.ndarray) -> Dict[str, float]:
    n = len(cp)
    ids = _SCRIPT_LUT[np.minimum(cp, 0xFFFF)]
    ids[cp > 0xFFFF] = 0                     # every listed range lies in the basic plane
    counts = np.bincount(ids, minlength=len(_SCRIPTS) + 1)
    return {name: int(counts[k + 1]) / n for k, (name, _) in enumerate(_SCRIPTS) if counts[k + 1]}


def _code_points(letters: str) -> np.ndarray:
    return np.frombuffer(letters.encode("utf-32-le", "surrogatepass"), dtype=np.uint32)


def script_shares(text: str) -> Dict[str, float]:
    """Share of the letters of ``text`` per script.  Runs at C speed (one regex pass to keep the letters, one table
    look-up and one histogram over their code points): a per-character Python loop held the interpreter lock for
    ~2.5 us per character, 50 ms for a 20 k-character document, in a service whose event loop shares that lock."""
    letters = "".join(_ALPHA.findall(text))
    return _shares_of(_code_points(letters)) if letters else {}


_BONUS = {lang: np.array([ord(c) for c in chars], dtype=np.int64) for lang, chars in _LETTERS.items()}
_BONUS_TOP = max(int(v.max()) for v in _BONUS.values()) + 1
# language-specific marks that are not letters (the regex pass drops them) are counted on the text itself
_BONUS_MARKS = {lang: [c for c in chars if not c.isalpha()] for lang, chars in _LETTERS.items()}


def detect(text: str, max_chars: int = 4000) -> Tuple[str, float]:
    """-> (language or script code, confidence in [0, 1]).  ``und`` when there is too little text to tell."""
    low = text[:max_chars].lower()
    toks = _TOKEN.findall(low)               # the one pass over the text: words, and joined, the letters
    letters = "".join(toks)
    if not letters:
        return "und", 0.0
    cp = _code_points(letters)
    shares = _shares_of(cp)
    if not shares:
        return "und", 0.0
    script, share = max(shares.items(), key=lambda kv: kv[1])
    if script != "latin":
        if script in ("kana",) or (script == "han" and shares.get("kana", 0.0) > 0.05):
            return "ja", share
        return {"han": "zh", "hangul": "ko"}.get(script, script), share
    if len(toks) < 3:
        return "und", 0.0
    scores: Dict[str, float] = {}
    seen = Counter(toks)
    per_letter = np.bincount(cp[cp < _BONUS_TOP].astype(np.int64), minlength=_BONUS_TOP)
    for lang, words in _SETS.items():
        hits = sum(seen[w] for w in words if w in seen)
        bonus = 0
        if lang in _BONUS:
            bonus = int(per_letter[_BONUS[lang]].sum()) + sum(low.count(c) for c in _BONUS_MARKS[lang])
        scores[lang] = hits + 0.5 * min(bonus, 0.2 * len(toks))
    best = max(scores, key=scores.get)
    ranked = sorted(scores.values(), reverse=True)
    if ranked[0] < 3:
        return "und", 0.0
    margin = (ranked[0] - ranked[1]) / ranked[0] if ranked[0] else 0.0
    return best, round(min(1.0, 0.5 + 0.5 * margin) * min(1.0, ranked[0] / (0.15 * len(toks) + 1)), 3)
