"""Native byte-level BPE vocabulary (csrc/runtime/bpe_tokenizer.cpp) against the `tokenizers` wheel the
reference depends on (requirements.txt:293): a ByteLevel BPE trained in-test (no files are reachable offline)."""
import json

import numpy as np
import pytest

from distributed_llms_amd.runtime.tokenizer import ByteLevelVocab, printable

CORPUS = [
    "the quick brown fox jumps over the lazy dog 12345 times",
    "Invoice No. 2024-17 total: $1,250.00 (paid) -- thank you!",
    "wave front matrix core tile shared memory gather scatter reduce",
    "naïve café résumé Zürich 東京 straße",
    "a  b   c\td\ne  multiple   spaces and\ttabs",
] * 8


def test_raw_byte_vocab_roundtrip_and_specials():
    v = ByteLevelVocab()
    assert (v.bos, v.pad, v.eos, v.unk, v.size) == (0, 1, 2, 3, 260)
    for text in CORPUS[:5]:
        ids = v.encode(text)
        assert ids == [4 + b for b in text.encode()]
        assert v.decode(ids) == text
    rows = np.full((3, 12), v.pad, dtype=np.int32)
    rows[0, :4] = [4 + ord("a"), 4 + ord("b"), v.eos, 4 + ord("z")]      # stops at eos
    rows[1, :3] = [v.bos, 4 + 0xC3, 4 + 0xA9]                            # bos skipped; two bytes = one code point
    rows[2, :2] = [9999, -5]                                             # out-of-range ids are dropped
    assert v.decode_batch(rows) == ["ab", "é", ""]
    assert v.decode_batch(rows, stop_at_eos=False)[0] == "abz"


def test_printable_alphabet_matches_gpt2_table():
    assert printable(b" ") == "Ġ" and printable(b"\n") == "Ċ" and printable(b"A~") == "A~"
    assert printable(bytes([0xAD])) == "Ń" and printable(bytes([0xFF])) == "ÿ"
    assert len({printable(bytes([b])) for b in range(256)}) == 256


def _train(tmp_path):
    tokenizers = pytest.importorskip("tokenizers")
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tr = trainers.BpeTrainer(vocab_size=420, special_tokens=["<s>", "<pad>", "</s>", "<unk>"],
                             initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(CORPUS, tr)
    tok.model.save(str(tmp_path))
    return tok


def test_loaded_vocab_matches_tokenizers_wheel(tmp_path):
    tok = _train(tmp_path)
    v = ByteLevelVocab.from_files(tmp_path / "vocab.json", tmp_path / "merges.txt")
    assert v.size == tok.get_vocab_size()
    assert (v.bos, v.pad, v.eos, v.unk) == (0, 1, 2, 3)
    for text in CORPUS[:5] + ["unseen words: kernel <42> {x=3}", "", "x"]:
        want = tok.encode(text).ids
        assert v.encode(text) == want, text
        assert v.decode(want) == text
    batch = [tok.encode(t).ids for t in CORPUS[:4]]
    L = max(map(len, batch))
    arr = np.array([[v.bos] + b + [v.eos] + [v.pad] * (L - len(b)) for b in batch], dtype=np.int32)
    assert v.decode_batch(arr) == CORPUS[:4]


def test_bad_vocab_is_rejected():
    with pytest.raises(ValueError):
        ByteLevelVocab.from_bytes(b"[1, 2, 3]")
    with pytest.raises(ValueError):
        ByteLevelVocab.from_bytes(b'{"a": "b"}')
    v = ByteLevelVocab.from_bytes(json.dumps({"<s>": 0, "</s>": 1, "a": 2, "Ġ": 3, "Ġa": 4}).encode(),
                                  "#version: 0.2\nĠ a\n".encode())
    assert v.encode("a a") == [2, 4] and v.decode([2, 4, 1, 2]) == "a a"
