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


def write_config(path: Path, weights_dir: Path, recognizer: str = "crnn") -> dict:
    cfg = {"weights-dir": str(weights_dir), "recognizer": recognizer, "device-mode": "gfx950",
           "det": {"limit-side": 1344, "bin-thresh": 0.3, "box-thresh": 0.6, "unclip-ratio": 1.5},
           "rec": {"height": 48, "charset": "ascii-printable"}}
    path.write_text(json.dumps(cfg, indent=2))
    return cfg


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--models-dir", default=os.path.expanduser("~/ocr-models"))
    ap.add_argument("--weights-dir", default=None, help="output directory (default: <models-dir>/weights)")
    ap.add_argument("--config", default=os.path.expanduser("~/ocr-engine.json"))
    ap.add_argument("--skip-download", action="store_true")
    args = ap.parse_args(argv)
    models_dir = Path(args.models_dir)
    weights_dir = Path(args.weights_dir) if args.weights_dir else models_dir / "weights"
    models_dir.mkdir(parents=True, exist_ok=True)
    if not args.skip_download:
        try:
            download(DEFAULT_REPOS, models_dir)
        except Exception as e:  # noqa: BLE001
            print(f"download failed ({e}); continuing with whatever is already in {models_dir}", file=sys.stderr)
    rep = convert_matching(models_dir, weights_dir)
    cfg = write_config(Path(args.config), weights_dir)
    print(json.dumps({"converted": rep, "config": args.config, "weights_dir": cfg["weights-dir"]}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
