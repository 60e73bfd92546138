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
