"""Model-download tool: conversion + config steps on locally generated safetensors (no network here)."""
import json

import pytest
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
    # letters outside the basic plane belong to no listed script but still count as letters; digits and marks do not
    shares = script_shares("ab \U00020000\U00020001 12 !?")
    assert shares == {"latin": 0.5}
    assert script_shares("漢字かな") == {"kana": 0.5, "han": 0.5}
    # language-specific marks that are not letters still vote (the inverted question mark for Spanish)
    assert detect("¿Que hora es? ¿Como te llamas? ¿Donde vive el medico de la familia de tu amigo?")[0] == "es"


def test_document_result_carries_the_language():
    from distributed_llms_amd.pdf import synth
    from distributed_llms_amd.runtime.document import DocumentPipeline
    from distributed_llms_amd.runtime.workers import build_engine

    lines = ["Der Vertrag ist von den Parteien nicht mehr zu ändern und er wird mit dem Datum der Unterschrift gültig.",
             "Die Kosten für das Verfahren werden von dem Kläger getragen, aber nur bis zu einem Betrag von 500 Euro."]
    eng = build_engine("cpu", {"model": "tiny", "cfg": dict(det_limit_side=160, rec_max_width=256, rec_width_quantum=64)})
    doc = DocumentPipeline(eng, decode_threads=1).process_pdf(synth.make_text_pdf([lines]), "lang")
    assert doc.mode == "txt" and doc.language == "de" and doc.language_confidence > 0.5


def _tiny_hf_trocr():
    transformers = pytest.importorskip("transformers")
    from transformers import TrOCRConfig as HfDec
    from transformers import VisionEncoderDecoderConfig, VisionEncoderDecoderModel, ViTConfig

    torch.manual_seed(0)
    enc = ViTConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, image_size=64,
                    patch_size=16, qkv_bias=False)
    dec = HfDec(vocab_size=272, d_model=128, decoder_layers=2, decoder_attention_heads=2, decoder_ffn_dim=256,
                max_position_embeddings=64, cross_attention_hidden_size=128, layernorm_embedding=True, scale_embedding=False,
                use_learned_position_embeddings=True, activation_function="gelu")
    cfg = VisionEncoderDecoderConfig.from_encoder_decoder_configs(enc, dec)
    m = VisionEncoderDecoderModel(cfg).eval()
    with torch.no_grad():                      # default init leaves LayerNorms at identity: make every tensor count
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return m, {"encoder": {"num_attention_heads": 2}, "decoder": {"decoder_attention_heads": 2, "activation_function": "gelu",
                                                                   "scale_embedding": False}}


@pytest.mark.parametrize("naming", ["current", "published"])
def test_hf_trocr_checkpoint_converts_with_identical_logits(naming):
    """A ViT + TrOCR VisionEncoderDecoder state dict (random weights, the architecture of microsoft/trocr-base-*)
    through tools.download_models.convert_hf_trocr: encoder memory and teacher-forced logits equal transformers' own,
    under both parameter namings found in the wild."""
    from distributed_llms_amd.tools.download_models import convert_hf_trocr

    hf, hc = _tiny_hf_trocr()
    sd = {k: v.clone() for k, v in hf.state_dict().items()}
    if naming == "published":
        ren = {}
        for k, v in sd.items():
            k2 = k
            if k.startswith("encoder.layers."):
                i = k.split(".")[2]
                rest = k.split(".", 3)[3]
                rest = (rest.replace("attention.q_proj", "attention.attention.query").replace("attention.k_proj", "attention.attention.key")
                        .replace("attention.v_proj", "attention.attention.value").replace("attention.o_proj", "attention.output.dense")
                        .replace("mlp.fc1", "intermediate.dense").replace("mlp.fc2", "output.dense"))
                k2 = f"encoder.encoder.layer.{i}.{rest}"
            ren[k2] = v
        sd = ren
        assert any(k.startswith("encoder.encoder.layer.0.attention.attention.query") for k in sd)
    ours, cfg = convert_hf_trocr(sd, hc)
    ours.eval()
    assert cfg.post_ln and cfg.embed_ln and cfg.pos_offset == 2 and not cfg.final_ln and cfg.image_size == 64
    x = torch.randn(3, 3, 64, 64)
    toks = torch.randint(3, 272, (3, 9))
    toks[:, 0] = 0
    with torch.no_grad():
        mem_hf = hf.encoder(pixel_values=x).last_hidden_state
        mem = ours.encode_reference(x)
        torch.testing.assert_close(mem, mem_hf, rtol=1e-4, atol=1e-4)
        logits_hf = hf(pixel_values=x, decoder_input_ids=toks).logits
        logits = ours.decode_logits_reference(mem, toks)
        torch.testing.assert_close(logits, logits_hf, rtol=1e-4, atol=2e-4)
        # greedy decoding agrees token for token
        gen_hf = hf.generate(pixel_values=x, max_new_tokens=8, do_sample=False, decoder_start_token_id=0, eos_token_id=None,
                             pad_token_id=1)
        assert torch.equal(ours.generate_reference(x, 8), gen_hf[:, 1:9])
    # the kernel path (CPU oracles of the HIP ops here) follows the same variant switches
    xi = torch.zeros(3, 64, 64, 8, dtype=torch.bfloat16)
    xi[..., :3] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    t_k, _ = ours.generate(xi, max_len=8, use_graph=False, eos_check_every=0)
    agree = (t_k.long() == gen_hf[:, 1:9]).float().mean().item()
    assert agree >= 0.8, agree                  # bf16 weights / activations: near-ties may flip


def test_trocr_snapshot_directory_converts_and_the_engine_builds_from_it(tmp_path):
    """What ``download_models --trocr`` leaves behind after the download: config.json + model.safetensors of a
    VisionEncoderDecoder snapshot -> trocr.safetensors + trocr_config.json, which ``build_engine`` turns into the
    engine's recogniser."""
    from safetensors.torch import save_file

    from distributed_llms_amd.runtime.workers import build_engine

    hf, hc = _tiny_hf_trocr()
    snap = tmp_path / "models" / "microsoft__trocr-tiny"
    snap.mkdir(parents=True)
    save_file({k: v.clone().contiguous() for k, v in hf.state_dict().items()}, str(snap / "model.safetensors"))
    (snap / "config.json").write_text(json.dumps({"model_type": "vision-encoder-decoder", **hc}))
    out = tmp_path / "weights"
    assert dm.convert_trocr_snapshots(tmp_path / "models", out) == [str(snap)]
    cfg = json.loads((out / "trocr_config.json").read_text())
    assert cfg["post_ln"] and cfg["pos_offset"] == 2 and cfg["image_size"] == 64 and cfg["vocab"] == 272
    # detector / CRNN files are not needed for this check: give the loader the shipped ones
    import shutil
    from pathlib import Path

    shipped = Path(__file__).resolve().parents[1] / "weights"
    for f in ("dbnet.safetensors", "crnn.safetensors"):
        if (shipped / f).exists():
            shutil.copy(shipped / f, out / f)
    if not (out / "dbnet.safetensors").exists():
        pytest.skip("no shipped detector weights")
    eng = build_engine("cpu", {"model": "full", "recognizer": "trocr", "weights_dir": str(out), "cfg": dict(max_batch_pages=1)})
    assert eng.rec_kind == "trocr" and eng.rec.cfg.post_ln and eng.rec.cfg.enc_layers == 2
    got = eng.rec.state_dict()["lm_head.weight"].float()
    assert torch.allclose(got, hf.state_dict()["decoder.output_projection.weight"].to(torch.bfloat16).float())
