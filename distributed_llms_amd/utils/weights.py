"""Model weight files: safetensors in, safetensors out.

The reference downloads third-party checkpoints at image-build time
(services/ocr_service/download_models_hf.py:36-51).  There is no network here, so
weights are either random-init or produced by ``train/`` and loaded from
``OCR_WEIGHTS_DIR`` (``dbnet.safetensors`` / ``crnn.safetensors`` / ``trocr.safetensors``).
Only safetensors is read - never pickles."""
from __future__ import annotations

import os
from typing import Dict, List

import torch


def save_model_weights(model: torch.nn.Module, path: str) -> None:
    from safetensors.torch import save_file

    sd = {k: v.detach().cpu().contiguous().clone() for k, v in model.state_dict().items()}
    tmp = f"{path}.tmp{os.getpid()}"
    save_file(sd, tmp)
    os.replace(tmp, path)


def load_model_weights(model: torch.nn.Module, path: str, strict: bool = True) -> List[str]:
    """Load a safetensors state dict.  Returns the list of keys that were missing in the file."""
    from safetensors.torch import load_file

    if not os.path.exists(path):
        if strict:
            raise FileNotFoundError(path)
        return list(model.state_dict().keys())
    sd: Dict[str, torch.Tensor] = load_file(path)
    res = model.load_state_dict(sd, strict=strict)
    return list(res.missing_keys)
