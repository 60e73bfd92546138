"""Model-download tool: conversion + config steps on locally generated safetensors (no network here)."""
import json

import torch

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.tools import download_models as dm
from distributed_llms_amd.utils.weights import load_model_weights, save_model_weights


def test_convert_matching_and_config(tmp_path):
    torch.manual_seed(1)
    src = CRNN()
    (tmp_path / "models" / "snap").mkdir(parents=True)
    save_model_weights(src, str(tmp_path / "models" / "snap" / "rec.safetensors"))
    rc = dm.main(["--models-dir", str(tmp_path / "models"), "--config", str(tmp_path / "cfg.json"), "--skip-download"])
    assert rc == 0
    cfg = json.loads((tmp_path / "cfg.json").read_text())
    assert cfg["recognizer"] == "crnn" and cfg["det"]["unclip-ratio"] == 1.5
    dst = CRNN()
    missing = load_model_weights(dst, str(tmp_path / "models" / "weights" / "crnn.safetensors"))
    assert missing == []
    for a, b in zip(src.state_dict().values(), dst.state_dict().values()):
        assert torch.equal(a, b)
    assert not (tmp_path / "models" / "weights" / "dbnet.safetensors").exists()   # nothing matched the detector


def test_language_identification_by_script_and_function_words():
    from distributed_llms_amd.utils.langid import detect, script_shares

    samples = {
        "en": "The quick brown fox jumps over the lazy dog and it was not there when we looked for it in the garden.",
        "de": "Der schnelle braune Fuchs springt über den faulen Hund und er ist nicht mehr in dem Garten zu sehen.",
        "fr": "Le renard brun rapide saute par-dessus le chien paresseux et il n'est pas dans le jardin avec nous.",
        "es": "El rápido zorro marrón salta sobre el perro perezoso y no está en el jardín con los otros animales.",
        "it": "La volpe marrone veloce salta sopra il cane pigro e non è più nel giardino con gli altri animali.",
        "pt": "A rápida raposa marrom pula sobre o cão preguiçoso e não está mais no jardim com os outros animais.",
        "nl": "De snelle bruine vos springt over de luie hond en hij is niet meer in de tuin met de andere dieren.",
        "cyrillic": "Быстрая коричневая лиса перепрыгивает через ленивую собаку.",
        "zh": "敏捷的棕色狐狸跳过了懒狗。",
        "ja": "すばやい茶色のキツネはのろまな犬を飛び越える。",
    }
    for want, text in samples.items():
        got, conf = detect(text)
        assert got == want and conf > 0.5, (want, got, conf)
    assert detect("42 17 3.5")[0] == "und" and detect("")[0] == "und" and detect("ok go")[0] == "und"
    assert script_shares("abc абв")["latin"] == 0.5


def test_document_result_carries_the_language():
    from distributed_llms_amd.pdf import synth
    from distributed_llms_amd.runtime.document import DocumentPipeline
    from distributed_llms_amd.runtime.workers import build_engine

    lines = ["Der Vertrag ist von den Parteien nicht mehr zu ändern und er wird mit dem Datum der Unterschrift gültig.",
             "Die Kosten für das Verfahren werden von dem Kläger getragen, aber nur bis zu einem Betrag von 500 Euro."]
    eng = build_engine("cpu", {"model": "tiny", "cfg": dict(det_limit_side=160, rec_max_width=256, rec_width_quantum=64)})
    doc = DocumentPipeline(eng, decode_threads=1).process_pdf(synth.make_text_pdf([lines]), "lang")
    assert doc.mode == "txt" and doc.language == "de" and doc.language_confidence > 0.5
