"""The recognizer can be taught: overfit a tiny set of strings with the CTC trainer, then read the same
lines back through the *inference* path (NHWC bf16 fused convs -> LSTM -> CTC greedy decode)."""
import numpy as np
import torch

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.train.trainer import CTCTrainer, TrainConfig


def test_crnn_overfits_and_inference_path_reads_it_back():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    cfg = TrainConfig(batch_size=8, width=128, lr=3e-3, vocab=4, max_chars=5, font_px=30, warmup_steps=10)
    tr = CTCTrainer(CRNN(hidden=64, widths=(16, 32, 32, 32, 64, 64)), cfg, device="cpu")
    hist = tr.fit(500)
    assert hist[-1]["loss"] < 0.2 < hist[0]["loss"]
    acc = tr.evaluate(2)
    assert acc["exact"] >= 0.8, acc
    # inference path on the same kind of samples
    model = tr.model.eval().prepare()
    x, il, tg, tl, texts = tr.data.batch(123456, 8)
    nhwc = torch.zeros(8, 48, 128, 8)
    nhwc[..., :3] = x.permute(0, 2, 3, 1)
    ids, ln, conf = model.infer(nhwc.to(torch.bfloat16), il.to(torch.int32))
    preds = model.decode_strings(ids, ln)
    want = ["".join(c for c in t if c in tr.data.idx) for t in texts]
    assert sum(p == w for p, w in zip(preds, want)) >= 6, list(zip(preds, want))
    assert float(conf.min()) > 0.5
