"""Per-layer time of every in-repo conv / small-M linear launch in one bench document (events around each launch;
hipGraphs off, single document, so the numbers are isolated kernel times, not pipelined wall time)."""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from distributed_llms_amd.models.crnn import CRNN
from distributed_llms_amd.models.dbnet import DBNet
from distributed_llms_amd.ops import conv
from distributed_llms_amd.pdf import synth
from distributed_llms_amd.runtime.document import DocumentPipeline
from distributed_llms_amd.runtime.engine import EngineConfig, OcrEngine
from distributed_llms_amd.utils.weights import load_model_weights

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
det, rec = DBNet(), CRNN()
load_model_weights(det, os.path.join(root, "weights/dbnet.safetensors"), strict=False)
load_model_weights(rec, os.path.join(root, "weights/crnn.safetensors"), strict=False)
eng = OcrEngine(torch.device("cuda:0"), det, rec, EngineConfig(max_batch_pages=25, use_hip_graphs=False), seed=1234)
pipe = DocumentPipeline(eng)
rendered = [synth.render_page(i, size=synth.A4_300DPI, font_px=42, margin=220) for i in range(8)]
pages = [rendered[i % 8][0] for i in range(100)]
enc = [synth.encode_page_jpeg(rendered[i][0]) for i in range(8)]
pdf = synth.make_scanned_pdf(pages, codec="jpeg", encoded=[enc[i % 8] for i in range(100)])
for _ in range(2):
    pipe.process_pdf(pdf, "w")
torch.cuda.synchronize()
conv.PROFILE_SINK = []
pipe.process_pdf(pdf, "p")
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for key, e0, e1 in conv.PROFILE_SINK:
    a = agg[key]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total conv/linear launch time for 100 pages: {tot:.1f} ms in {sum(v[0] for v in agg.values())} launches")
print(f"{'M':>9s} {'N':>5s} {'K':>5s} {'kind':28s} {'calls':>5s} {'ms':>8s} {'TF':>7s} {'GB/s(in+out)':>12s}")
for (M, N, K, kind), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tf = 2.0 * M * N * K * n / (ms * 1e-3) / 1e12
    taps = int(kind[0]) * int(kind[2]) if kind[0].isdigit() and kind[2].isdigit() else 1
    gbs = (M * N * 2 + M * K * 2 / max(1, taps)) * n / (ms * 1e-3) / 1e9
    print(f"{M:9d} {N:5d} {K:5d} {kind:28s} {n:5d} {ms:8.2f} {tf:7.1f} {gbs:12.0f}")
