"""Fetch pretrained OCR checkpoints and write the engine's model config.

Counterpart of the reference's ``download_models_hf.py`` (services/ocr_service/download_models_hf.py:9-74), which
pulls ``opendatalab/PDF-Extract-Kit-1.0`` (layout / formula / OCR models) and ``hantian/layoutreader`` with
``huggingface_hub.snapshot_download`` and then patches ``~/magic-pdf.json``.  This engine has its own model
definitions, so the tool

1. downloads the requested Hugging Face snapshots (same call, same ``allow_patterns`` idea) into
   ``--models-dir``;
2. converts any ``*.safetensors`` / state-dict whose tensor names and shapes match ``DBNet`` / ``CRNN`` /
   ``TrOCR`` into ``dbnet.safetensors`` / ``crnn.safetensors`` / ``trocr.safetensors`` (only safetensors is
   read - pickled checkpoints are refused);
3. writes ``ocr-engine.json`` (the analogue of ``magic-pdf.json``): weights dir, recognizer, thresholds.

The build environment of this repository has no network, so step 1 could not be exercised there; steps 2-3
are covered by tests/test_tools.py with locally generated files.  Point ``OCR_WEIGHTS_DIR`` at the output."""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path
from typing import Dict, List, Optional

# an encoder-decoder recogniser with public weights in the layout ``convert_hf_trocr`` understands (``--trocr``)
TROCR_REPO = ("microsoft/trocr-base-printed", ["config.json", "*.safetensors", "vocab.json", "merges.txt"])

DEFAULT_REPOS = {
    # repo id -> allow_patterns (the reference's list, download_models_hf.py:36-44)
    "opendatalab/PDF-Extract-Kit-1.0": ["models/OCR/paddleocr_torch/*", "models/Layout/YOLO/*", "models/MFD/YOLO/*",
                                        "models/MFR/unimernet_hf_small_2503/*"],
    "hantian/layoutreader": ["*.json", "*.safetensors"],
}


def download(repos: Dict[str, List[str]], models_dir: Path) -> Dict[str, str]:
    from huggingface_hub import snapshot_download

    out = {}
    for repo, patterns in repos.items():
        path = snapshot_download(repo, allow_patterns=patterns, local_dir=str(models_dir / repo.replace("/", "__")))
        out[repo] = path
        print(f"downloaded {repo} -> {path}")
    return out


def convert_matching(src_dir: Path, dst_dir: Path) -> Dict[str, int]:
    """Copy every tensor whose name and shape match one of our models into that model's weight file."""
    import torch
    from safetensors.torch import load_file

    from ..models.crnn import CRNN
    from ..models.dbnet import DBNet
    from ..utils.weights import save_model_weights

    dst_dir.mkdir(parents=True, exist_ok=True)
    pool: Dict[str, "torch.Tensor"] = {}
    for f in sorted(src_dir.rglob("*.safetensors")):
        try:
            pool.update(load_file(str(f)))
        except Exception as e:  # noqa: BLE001
            print(f"skipping {f}: {e}", file=sys.stderr)
    report = {}
    for name, model in (("dbnet", DBNet()), ("crnn", CRNN())):
        sd = model.state_dict()
        hit = {k: pool[k] for k in sd if k in pool and tuple(pool[k].shape) == tuple(sd[k].shape)}
        report[name] = len(hit)
        if hit:
            model.load_state_dict({**sd, **hit})
            save_model_weights(model, str(dst_dir / f"{name}.safetensors"))
            print(f"{name}: {len(hit)}/{len(sd)} tensors taken from the download")
    return report


def convert_hf_trocr(state: Dict[str, "torch.Tensor"], hf_config: Optional[dict] = None):
    """A ``VisionEncoderDecoderModel`` state dict (ViT encoder + TrOCR decoder, e.g. microsoft/trocr-base-printed)
    -> (``models.trocr.TrOCR`` with those weights, its config).

    Both parameter namings are understood: the one the published checkpoints carry
    (``encoder.encoder.layer.N.attention.attention.query`` ...) and the one recent ``transformers`` releases write
    (``encoder.layers.N.attention.q_proj`` ...).  Shapes give the geometry; ``hf_config`` (the checkpoint's
    ``config.json`` as a dict) supplies what shapes cannot: head counts, activation, ``scale_embedding``.
    tests/test_tools.py checks logits against ``transformers`` itself on a randomly initialised model."""
    import torch

    from ..models.trocr import TrOCR, hf_trocr_config

    sd = {k: v for k, v in state.items()}
    old = any(k.startswith("encoder.encoder.layer.") for k in sd)

    def enc(i: int, what: str) -> str:
        if old:
            return {"q": f"encoder.encoder.layer.{i}.attention.attention.query", "k": f"encoder.encoder.layer.{i}.attention.attention.key",
                    "v": f"encoder.encoder.layer.{i}.attention.attention.value", "o": f"encoder.encoder.layer.{i}.attention.output.dense",
                    "ln1": f"encoder.encoder.layer.{i}.layernorm_before", "ln2": f"encoder.encoder.layer.{i}.layernorm_after",
                    "fc1": f"encoder.encoder.layer.{i}.intermediate.dense", "fc2": f"encoder.encoder.layer.{i}.output.dense"}[what]
        return {"q": f"encoder.layers.{i}.attention.q_proj", "k": f"encoder.layers.{i}.attention.k_proj",
                "v": f"encoder.layers.{i}.attention.v_proj", "o": f"encoder.layers.{i}.attention.o_proj",
                "ln1": f"encoder.layers.{i}.layernorm_before", "ln2": f"encoder.layers.{i}.layernorm_after",
                "fc1": f"encoder.layers.{i}.mlp.fc1", "fc2": f"encoder.layers.{i}.mlp.fc2"}[what]

    n_enc = 0
    while enc(n_enc, "q") + ".weight" in sd:
        n_enc += 1
    n_dec = 0
    while f"decoder.model.decoder.layers.{n_dec}.fc1.weight" in sd:
        n_dec += 1
    if n_enc == 0 or n_dec == 0:
        raise ValueError("not a ViT + TrOCR VisionEncoderDecoder state dict")
    pw = sd["encoder.embeddings.patch_embeddings.projection.weight"]
    enc_dim, patch = int(pw.shape[0]), int(pw.shape[2])
    n_tok = int(sd["encoder.embeddings.position_embeddings"].shape[1])
    if "encoder.embeddings.distillation_token" in sd:
        raise ValueError("DeiT encoders (distillation token: the trocr-small checkpoints) are not supported")
    grid = int(round((n_tok - 1) ** 0.5))
    dec_dim = int(sd["decoder.model.decoder.embed_tokens.weight"].shape[1])
    vocab = int(sd["decoder.model.decoder.embed_tokens.weight"].shape[0])
    hc = hf_config or {}
    ec, dc = hc.get("encoder", {}), hc.get("decoder", {})
    cfg = hf_trocr_config(
        image_size=grid * patch, patch=patch, enc_dim=enc_dim, enc_layers=n_enc, enc_heads=int(ec.get("num_attention_heads", enc_dim // 64)),
        enc_mlp=int(sd[enc(0, "fc1") + ".weight"].shape[0]), dec_dim=dec_dim, dec_layers=n_dec,
        dec_heads=int(dc.get("decoder_attention_heads", dec_dim // 64)),
        dec_ffn=int(sd["decoder.model.decoder.layers.0.fc1.weight"].shape[0]), vocab=vocab,
        max_len=int(sd["decoder.model.decoder.embed_positions.weight"].shape[0]) - 2,
        act=str(dc.get("activation_function", "gelu")), scale_embedding=bool(dc.get("scale_embedding", False)),
        layernorm_embedding="decoder.model.decoder.layernorm_embedding.weight" in sd)
    if cfg.enc_dim != 64 * cfg.enc_heads or cfg.dec_dim != 64 * cfg.dec_heads:
        raise ValueError("only 64-wide attention heads are supported by the attention kernels")
    m = TrOCR(cfg)

    def lin(dst, name, bias_dim=None):
        dst.weight.copy_(sd[name + ".weight"])
        if dst.bias is not None:
            dst.bias.copy_(sd[name + ".bias"]) if name + ".bias" in sd else dst.bias.zero_()

    def fused(dst, names):
        dst.weight.copy_(torch.cat([sd[n + ".weight"] for n in names], 0))
        dst.bias.copy_(torch.cat([sd[n + ".bias"] if n + ".bias" in sd else torch.zeros(sd[n + ".weight"].shape[0]) for n in names], 0))

    with torch.no_grad():
        m.patch_embed.weight.copy_(pw)
        m.patch_embed.bias.copy_(sd["encoder.embeddings.patch_embeddings.projection.bias"])
        m.cls.copy_(sd["encoder.embeddings.cls_token"])
        m.enc_pos.copy_(sd["encoder.embeddings.position_embeddings"])
        for i, l in enumerate(m.enc):
            fused(l.qkv, [enc(i, "q"), enc(i, "k"), enc(i, "v")])
            lin(l.proj, enc(i, "o"))
            lin(l.ln1, enc(i, "ln1"))
            lin(l.ln2, enc(i, "ln2"))
            lin(l.fc1, enc(i, "fc1"))
            lin(l.fc2, enc(i, "fc2"))
        lin(m.enc_ln, "encoder.layernorm")
        d = "decoder.model.decoder."
        m.tok.weight.copy_(sd[d + "embed_tokens.weight"])
        m.pos.weight.copy_(sd[d + "embed_positions.weight"])
        if m.emb_ln is not None:
            lin(m.emb_ln, d + "layernorm_embedding")
        for i, l in enumerate(m.dec):
            p = f"{d}layers.{i}."
            fused(l.qkv, [p + "self_attn.q_proj", p + "self_attn.k_proj", p + "self_attn.v_proj"])
            lin(l.proj, p + "self_attn.out_proj")
            lin(l.ln1, p + "self_attn_layer_norm")
            lin(l.xq, p + "encoder_attn.q_proj")
            fused(l.xkv, [p + "encoder_attn.k_proj", p + "encoder_attn.v_proj"])
            lin(l.xproj, p + "encoder_attn.out_proj")
            lin(l.ln2, p + "encoder_attn_layer_norm")
            lin(l.fc1, p + "fc1")
            lin(l.fc2, p + "fc2")
            lin(l.ln3, p + "final_layer_norm")
        # checkpoints with tied embeddings store the matrix once
        m.lm_head.weight.copy_(sd.get("decoder.output_projection.weight", sd[d + "embed_tokens.weight"]))
    return m, cfg


def convert_trocr_snapshots(src_dir: Path, dst_dir: Path) -> List[str]:
    """Every VisionEncoderDecoder snapshot under ``src_dir`` (a ``config.json`` with that model type next to
    ``*.safetensors``) -> ``trocr.safetensors`` + ``trocr_config.json`` (+ vocab files) in ``dst_dir``; the engine
    builds the recogniser from that config (runtime/workers.py::build_engine, ``OCR_RECOGNIZER=trocr``)."""
    from dataclasses import asdict

    from safetensors.torch import load_file, save_file

    done = []
    for cj in sorted(src_dir.rglob("config.json")):
        try:
            hc = json.loads(cj.read_text())
        except ValueError:
            continue
        if hc.get("model_type") != "vision-encoder-decoder":
            continue
        state = {}
        for f in sorted(cj.parent.glob("*.safetensors")):
            state.update(load_file(str(f)))
        if not state:
            continue
        try:
            model, cfg = convert_hf_trocr(state, hc)
        except (ValueError, KeyError) as e:
            print(f"skipping {cj.parent}: {e}", file=sys.stderr)
            continue
        dst_dir.mkdir(parents=True, exist_ok=True)
        sd = {k: v.detach().to(torch_bf16()).contiguous() if v.is_floating_point() else v.detach().contiguous()
              for k, v in model.state_dict().items()}
        save_file(sd, str(dst_dir / "trocr.safetensors"))
        (dst_dir / "trocr_config.json").write_text(json.dumps(asdict(cfg), indent=1))
        for extra in ("vocab.json", "merges.txt"):
            if (cj.parent / extra).exists():
                (dst_dir / extra).write_bytes((cj.parent / extra).read_bytes())
        done.append(str(cj.parent))
    return done


def torch_bf16():
    import torch

    return torch.bfloat16


def write_config(path: Path, weights_dir: Path, recognizer: str = "crnn") -> dict:
    cfg = {"weights-dir": str(weights_dir), "recognizer": recognizer, "device-mode": "gfx950",
           "det": {"limit-side": 1344, "bin-thresh": 0.3, "box-thresh": 0.6, "unclip-ratio": 1.5},
           "rec": {"height": 48, "charset": "from-weights (classifier rows select a prefix of models/crnn.py::DEFAULT_CHARSET)"}}
    path.write_text(json.dumps(cfg, indent=2))
    return cfg


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--models-dir", default=os.path.expanduser("~/ocr-models"))
    ap.add_argument("--weights-dir", default=None, help="output directory (default: <models-dir>/weights)")
    ap.add_argument("--config", default=os.path.expanduser("~/ocr-engine.json"))
    ap.add_argument("--skip-download", action="store_true")
    ap.add_argument("--trocr", action="store_true", help=f"also fetch {TROCR_REPO[0]} and convert it for OCR_RECOGNIZER=trocr")
    args = ap.parse_args(argv)
    models_dir = Path(args.models_dir)
    weights_dir = Path(args.weights_dir) if args.weights_dir else models_dir / "weights"
    models_dir.mkdir(parents=True, exist_ok=True)
    if not args.skip_download:
        try:
            download({**DEFAULT_REPOS, **({TROCR_REPO[0]: TROCR_REPO[1]} if args.trocr else {})}, models_dir)
        except Exception as e:  # noqa: BLE001
            print(f"download failed ({e}); continuing with whatever is already in {models_dir}", file=sys.stderr)
    rep = convert_matching(models_dir, weights_dir)
    trocr = convert_trocr_snapshots(models_dir, weights_dir)
    cfg = write_config(Path(args.config), weights_dir, recognizer="trocr" if trocr else "crnn")
    print(json.dumps({"converted": rep, "trocr_snapshots": trocr, "config": args.config, "weights_dir": cfg["weights-dir"]}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
