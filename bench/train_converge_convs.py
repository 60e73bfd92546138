"""Short GPU convergence check of the CTC trainer: convolutions on the in-repo training kernels (ops/conv_train.py)
against the library's, same seed, same data."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.train.trainer import CTCTrainer, TrainConfig

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for own in (True, False):
    torch.manual_seed(0)
    tr = CTCTrainer(CRNN(), TrainConfig(batch_size=64, width=256, lr=2e-3, vocab=64, max_chars=10, warmup_steps=10,
                                        own_convs=own), "cuda:0")
    t0 = time.perf_counter()
    hist = tr.fit(steps)
    torch.cuda.synchronize()
    ev = tr.evaluate(2)
    print(json.dumps({"own_convs": own, "steps": steps, "s": round(time.perf_counter() - t0, 1),
                      "loss_first": round(hist[0]["loss"], 3), "loss_at_100": round(hist[min(99, steps - 1)]["loss"], 4),
                      "loss_last": round(hist[-1]["loss"], 4), **{k: round(v, 4) for k, v in ev.items()}}), flush=True)
