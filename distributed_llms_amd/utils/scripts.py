"""Cyrillic and Greek next to Latin in ONE recogniser: visual classes and script resolution.

The reference downloads one multilingual recogniser and lets its engine pick by document language
(services/ocr_service/download_models_hf.py:41, services/ocr_service/ocr_service.py:215-220).  A printed Cyrillic
``А`` or Greek ``Α`` is the same shape as a Latin ``A``: a classifier with a class per code point would have to guess
between identical glyphs.  So the recogniser's classes are *shapes*:

* letters of the three alphabets that share a shape share the Latin class (``HOMOGLYPHS``: 20 Cyrillic and 16 Greek
  code points);
* the letters only one alphabet has (``CYRILLIC_EXTRA``, ``GREEK_EXTRA``) are classes of their own, appended to the
  Latin character set - a model grown from Latin weights keeps its class indices.

``to_visual`` maps truth text to classes for training; ``resolve_scripts`` turns a decoded line back into code
points: a word that holds a Cyrillic-only (Greek-only) letter is Cyrillic (Greek) and its shared shapes become that
alphabet's letters; a word made of shared shapes only ("сорт" without the т, "о", "а") follows the majority of the
line.  Lines without a non-Latin class take a fast path and come back unchanged.
"""
from __future__ import annotations

from typing import Dict, Optional

# letters that no Latin letter looks like, in alphabet order
CYRILLIC_EXTRA = "БГДЖЗИЙЛПУФЦЧШЩЪЫЬЭЮЯ" "бвгджзийклмнптфцчшщъыьэюя" "№"
GREEK_EXTRA = "ΓΔΘΛΞΠΣΦΨΩ" "αβγδεζηθικλμνξπρστυφχψως" "άέήίύώ" "ΆΈΉΊΌΎΏ"

# code point -> the Latin character whose class it shares
_CYR_TO_LATIN = dict(zip("АВЕКМНОРСТХаеорсухЁё", "ABEKMHOPCTXaeopcyxËë"))
_GRK_TO_LATIN = dict(zip("ΑΒΕΖΗΙΚΜΝΟΡΤΥΧοό", "ABEZHIKMNOPTYXoó"))
HOMOGLYPHS: Dict[str, str] = {**_CYR_TO_LATIN, **_GRK_TO_LATIN}
_LATIN_TO_CYR = {v: k for k, v in _CYR_TO_LATIN.items()}
_LATIN_TO_GRK = {v: k for k, v in _GRK_TO_LATIN.items()}
# Near-shapes the recogniser may pick inside a word of the other alphabet, applied only to words already known to be
# Cyrillic / Greek from their unambiguous letters.  In italic cuts several of them are the very same glyph: Cyrillic
# и, п, т, г, д are drawn as u, n, m, r-like and g-like forms, Greek alpha as a, nu as v.
_LATIN_TO_CYR_SOFT = {**_LATIN_TO_CYR, "k": "к", "u": "и", "n": "п", "m": "т", "r": "г", "g": "д"}
_LATIN_TO_GRK_SOFT = {**_LATIN_TO_GRK, "v": "ν", "u": "υ", "k": "κ", "i": "ι", "a": "α", "n": "η", "p": "ρ", "w": "ω",
                      "t": "τ", "e": "ε", "x": "χ", "y": "γ", "á": "ά", "í": "ί", "é": "έ"}

_CYR_ONLY = frozenset(CYRILLIC_EXTRA) - {"№"}
_GRK_ONLY = frozenset(GREEK_EXTRA)
_SHARED = frozenset(HOMOGLYPHS.values())
_VISUAL = str.maketrans(HOMOGLYPHS)
_TO_CYR = str.maketrans(_LATIN_TO_CYR_SOFT)
_TO_GRK = str.maketrans(_LATIN_TO_GRK_SOFT)


def to_visual(text: str) -> str:
    """Truth text -> the recogniser's classes (shared shapes written as their Latin letter)."""
    return text if text.isascii() else text.translate(_VISUAL)


def _counts(word: str):
    cyr = grk = lat = 0
    for c in word:
        if c in _CYR_ONLY:
            cyr += 1
        elif c in _GRK_ONLY:
            grk += 1
        elif c.isalpha() and c not in _SHARED and c not in HOMOGLYPHS:
            lat += 1
    return cyr, grk, lat


def script_counts(text: str):
    """(letters only Cyrillic has, letters only Greek has, letters only Latin has) in ``text``."""
    return _counts(text)


def line_script(text: str) -> str:
    """``latin``, ``cyrillic`` or ``greek``: the alphabet most of the unambiguous letters of a decoded line belong to."""
    if text.isascii():
        return "latin"
    cyr, grk, lat = _counts(text)
    if cyr > max(grk, lat):
        return "cyrillic"
    if grk > max(cyr, lat):
        return "greek"
    return "latin"


# short English words that are spelled with twin letters only: left alone inside a Russian / Greek line
_SHORT_ENGLISH = frozenset(
    "a an am at be by do go he if in is it me my no of on or so to up us we an and any are ask but buy can car cut day "
    "eat end eye for get got guy her him his hot how its key man map may men met net new nor not now one our out own "
    "pay per put ran run saw say see set she sit six ten the too top try two use war was way who why win won yes yet you "
    "pm km cm mm ok tv pc api cpu gpu url pdf ocr".split())


def _short_foreign(word: str, dom: str) -> bool:
    """A word of at most three letters, in a line of the other alphabet, all of whose letters have a (near-)twin
    there: the italic "om", "u", "mo" of a Russian line are от, и, то.  Common short English words ("no", "one", "at") and
    longer words stay as read - an English word inside a Russian sentence is far more likely than five twins in a row."""
    table = _LATIN_TO_CYR_SOFT if dom == "cyrillic" else _LATIN_TO_GRK_SOFT
    letters = [c for c in word if c.isalpha()]
    if not 0 < len(letters) <= 3 or "".join(letters).lower() in _SHORT_ENGLISH:
        return False
    return all(c in table for c in letters)


# letters a digit can be taken for; inside a token that is otherwise digits and punctuation they are read as the digit
_DIGIT_TWINS = str.maketrans({"б": "6", "З": "3", "з": "3", "О": "0", "о": "0", "O": "0", "o": "0", "l": "1", "I": "1",
                              "i": "1", "Ь": "6", "ь": "6"})


def _numeric_token(word: str) -> bool:
    """At least two digits, and every letter in the token has a digit twin ("4б18.89", "2O24")."""
    digits = sum(c.isdigit() for c in word)
    letters = [c for c in word if c.isalpha()]
    return digits >= 2 and 0 < len(letters) <= digits and all(ord(c) in _DIGIT_TWINS for c in letters)


def resolve_scripts(text: str, hint: Optional[str] = None) -> str:
    """Decoded classes -> code points.  ``hint``: the script of the surrounding text (a page or document majority) for
    lines that consist of shared shapes only."""
    if text.isascii() and hint in (None, "latin"):
        return text
    dom = line_script(text)
    if dom == "latin" and hint in ("cyrillic", "greek") and _counts(text)[2] == 0:
        dom = hint
    if dom == "latin" and not (_CYR_ONLY.intersection(text) or _GRK_ONLY.intersection(text)):
        return text
    out = []
    for word in text.split(" "):
        if _numeric_token(word):
            out.append(word.translate(_DIGIT_TWINS))
            continue
        cyr, grk, lat = _counts(word)
        if cyr and cyr >= max(grk, lat):
            word = word.translate(_TO_CYR)
        elif grk and grk >= max(cyr, lat):
            word = word.translate(_TO_GRK)
        elif dom != "latin" and not (cyr or grk) and (lat == 0 or _short_foreign(word, dom)):
            word = word.translate(_TO_CYR if dom == "cyrillic" else _TO_GRK)
        out.append(word)
    return " ".join(out)


# ------------------------------------------------------------------------------------------------ vocabularies
# High-frequency and document words for synthetic pages; together they use every letter of the two alphabets.
WORDS_RU = (
    "и в не на я что он с как а то это по но все она так его же за от было бы у мы к вы из они для при ещё уже "
    "или если только когда может быть есть нет да где кто чем этот который свой наш ваш один два три год время "
    "дело жизнь день рука раз работа слово место вопрос сторона страна мир случай голова ребёнок сила конец вид "
    "система часть город отношение женщина деньги земля машина вода отец проблема час право решение дверь образ "
    "история власть закон война книга возможность результат область статья число компания народ группа развитие "
    "процесс условие средство начало государство уровень форма связь минута качество действие правительство "
    "документ страница таблица рисунок глава раздел приложение отчёт счёт сумма итого дата номер подпись договор "
    "общество директор бухгалтер заявление справка адрес телефон объект эксперт юрист цель язык щит съезд подъезд "
    "Москва Россия Петербург Фёдор Юрий Цветаева Чехов Шишкин Щедрин Эльбрус Яков Жуков Зоя Иван Йошкар Люба Пётр "
    "Ульяна Фома Харьков Гоголь Дон Байкал Владимир Ярославль Эрмитаж Южный Центр Чайковский Шаляпин Ъ Ы Ь "
    "№ 15 2024 г. 1 250,00 руб. 37% стр. 12 (а) [3] т. е. и т. д."
).split()
WORDS_EL = (
    "και το να η ο του της με που τα για από δεν θα την στο είναι σε τον των μου ένα μια αλλά αυτό σου στην όπως τι "
    "οι τους αν κι μας στη ή ότι πως είχε ήταν έχει πιο όλα μετά πολύ εκεί τώρα πριν όταν γιατί χρόνια ημέρα χρόνος "
    "άνθρωπος ζωή κόσμος πόλη χώρα νερό δουλειά λόγος μέρος πράγμα παιδί γυναίκα άντρας σπίτι χέρι μάτι φορά ώρα "
    "βιβλίο έγγραφο σελίδα πίνακας εικόνα κεφάλαιο ενότητα παράρτημα αναφορά σύνολο ποσό ημερομηνία αριθμός υπογραφή "
    "σύμβαση εταιρεία διευθυντής αίτηση διεύθυνση τηλέφωνο αποτέλεσμα σύστημα ανάπτυξη κυβέρνηση νόμος ιστορία ψυχή "
    "ξένος ζήτημα θέμα φως ψωμί ξύλο γλώσσα δρόμος ύψος ώμος ήλιος έργο ίδιος "
    "Αθήνα Ελλάδα Θεσσαλονίκη Γιώργος Δημήτρης Ζωή Ηλίας Θεός Ιωάννης Κώστας Λάρισα Μαρία Νίκος Ξάνθη Ολυμπία "
    "Πάτρα Ρόδος Σπάρτη Τρίκαλα Υπάτη Φώτης Χανιά Ψαρά Ωρωπός Βόλος Γ Δ Θ Λ Ξ Π Σ Φ Ψ Ω "
    "15 2024 1.250,00 € 37% σελ. 12 (α) [3] κ.λπ. π.χ."
).split()
