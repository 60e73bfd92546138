"""Short GPU convergence check of the CTC trainer (fused LSTM recurrences vs the eager nn.LSTM path)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.train.trainer import CTCTrainer, TrainConfig

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for fused in (True, False):
    torch.manual_seed(0)
    model = CRNN()
    model.fused_train_lstm = fused
    tr = CTCTrainer(model, TrainConfig(batch_size=64, width=256, lr=2e-3, vocab=64, max_chars=10, warmup_steps=10), "cuda:0")
    t0 = time.perf_counter()
    n = steps if fused else min(steps, 120)
    hist = tr.fit(n)
    torch.cuda.synchronize()
    ev = tr.evaluate(2)
    print(json.dumps({"fused_lstm": fused, "steps": n, "s": round(time.perf_counter() - t0, 1),
                      "loss_first": round(hist[0]["loss"], 3), "loss_at_120": round(hist[min(119, n - 1)]["loss"], 4),
                      "loss_last": round(hist[-1]["loss"], 4), **{k: round(v, 4) for k, v in ev.items()}}), flush=True)
