"""Cyrillic / Greek shape classes (utils/scripts.py) and the recogniser's multi-script character set."""
import torch

from distributed_llms_amd.models.crnn import CRNN, DEFAULT_CHARSET, MULTI_CHARSET
from distributed_llms_amd.utils import scripts as S


def test_every_vocabulary_word_survives_classes_and_back():
    cs = set(MULTI_CHARSET)
    for words in (S.WORDS_RU, S.WORDS_EL):
        for w in words:
            for form in (w, w.capitalize(), w.upper()):
                v = S.to_visual(form)
                assert all(c in cs for c in v), (form, v)
    # whole lines: words made of shared shapes only ("о", "а", "сор") follow the line
    for line in ("Привет, мир! Это тест № 5", "сорт о в а не то", "Η Αθήνα είναι η πρωτεύουσα της Ελλάδας", "ΟΤΕ και ΔΕΗ",
                 "Москва Moscow 2024 Worker", "Το σύστημα GPU και ο server", "Отчёт Фёдора за 2024 г."):
        assert S.resolve_scripts(S.to_visual(line)) == line, line


def test_latin_text_is_left_alone():
    for line in ("The cop saw a MAP", "Noël à l'hôtel: 12,50 €", "x=3 y<9 #42", ""):
        assert S.to_visual(line) == line and S.resolve_scripts(line) == line
    assert S.line_script("plain ascii") == "latin"


def test_italic_twins_are_resolved_by_context():
    """Italic Cyrillic draws и, п, т as u, n, m; italic Greek draws alpha as a and nu as v.  Inside a word (or, for
    very short words, a line) of that alphabet they are read as its letters; English words in the line stay."""
    assert S.resolve_scripts("Этo пo гpyппe om ниx u mo") == "Это по группе от них и то"
    assert S.resolve_scripts("Этo no one at гpyппe") == "Это no one at группе"          # short English words stay
    assert S.resolve_scripts("Пpивeт Worker and Queue") == "Привет Worker and Queue"
    assert S.resolve_scripts("Όταv τηv a και va") == "Όταν την α και να"
    # a digit read as its look-alike letter inside a number
    assert S.resolve_scripts("4бi8.89 pyб. и 2O24") == "4618.89 руб. и 2024"


def test_recogniser_grows_to_the_multi_script_set_and_reads_it_back():
    m = CRNN()
    assert m.charset == DEFAULT_CHARSET and not m.multi_script
    w0 = m.fc.weight.detach().clone()
    m.set_charset(MULTI_CHARSET)
    assert m.multi_script and m.fc.out_features == len(MULTI_CHARSET) + 1 == 258
    assert torch.equal(m.fc.weight[: w0.shape[0]], w0)                 # Latin classes keep their rows
    text = "Привет, мир № 5 Ελλάδα Straße"
    ids = m.encode_text(text)
    assert len(ids) == len(text)
    t = torch.tensor([ids], dtype=torch.int32)
    assert m.decode_strings(t, torch.tensor([len(ids)], dtype=torch.int32)) == [text]
    m2 = CRNN()
    m2.load_state_dict(m.state_dict())                                  # a 258-row checkpoint brings its character set
    assert m2.charset == MULTI_CHARSET


def test_page_majority_decides_lines_of_shared_letters():
    """"О НАС", "ТЕМА", a lone "а": every letter has a Latin twin, so the line cannot tell its alphabet; on a page whose
    telling lines are mostly Cyrillic it is read as Cyrillic, on a Latin page it stays as it is."""
    from distributed_llms_amd.runtime.engine import TextLine, _apply_page_script

    texts = ["О НАС", "Привет, мир и все", "Это тест системы", "Москва и область", "ТЕМА", "Worker and Queue", "2024", "а"]
    lines = [TextLine((0, 0, 1, 1), S.resolve_scripts(S.to_visual(t)), 1.0) for t in texts]
    assert lines[0].text == "O HAC" and lines[4].text == "TEMA"            # read alone they are Latin
    _apply_page_script(lines)
    assert [ln.text for ln in lines] == texts
    latin = [TextLine((0, 0, 1, 1), t, 1.0) for t in ["The cop saw", "A MAP", "Привет one", "HOME"]]
    _apply_page_script(latin)
    assert [ln.text for ln in latin] == ["The cop saw", "A MAP", "Привет one", "HOME"]
