"""Evaluate the shipped weights through the engine on the page trainer's held-out sets (what train_meta.json records):
``python bench/eval_weights.py [WEIGHTS_DIR] [inline]``."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.models.dbnet import DBNet
from distributed_llms_amd.train.page_trainer import evaluate
from distributed_llms_amd.utils.weights import load_model_weights

wd = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weights")
inline = len(sys.argv) > 2 and sys.argv[2] == "inline"      # the detector for formulas inside lines, and the engine cutting them out
det_file = "dbnet_inline.safetensors" if inline and os.path.exists(os.path.join(wd, "dbnet_inline.safetensors")) else "dbnet.safetensors"
dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
det, rec = DBNet().to(dev), CRNN().to(dev)
load_model_weights(det, os.path.join(wd, det_file), strict=False)
load_model_weights(rec, os.path.join(wd, "crnn.safetensors"))
ev = evaluate(det.eval(), rec.eval(), dev, list(range(900_000, 900_008)), formula_weights=os.path.join(wd, "formula.safetensors"),
              inline=inline)
print(json.dumps({k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in ev.items()}))
