#!/usr/bin/env python3
"""Every launch class of the headline step by (kind, FLOPs): launches per step, average duration inside the step, achieved TFLOP/s --
to find GEMMs that waste matrix-core time (tile padding, too few tiles for the chip, short K)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import bench, b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(10)
model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)
def step(i):
    return ts.step(ops.augment_smooth(x, 2, 100, "same", cut=i % 3, white_std=1.0, offset_std=0.2, seed=i), days, labels, nts - i % 3, lens)
for i in range(6): step(i)
torch.cuda.synchronize()
model._ws.profile(True)
NS = 4
for i in range(NS): step(6 + i)
recs = model._ws.profile_read()
model._ws.profile(False)
agg = collections.defaultdict(lambda: [0, 0.0])
for name, fl, n, sec in recs:
    k = (name, round(fl / 1e9, 2))
    agg[k][0] += 1; agg[k][1] += sec
print(f"{'kind':28s} {'GFLOP':>9s} {'launches/step':>13s} {'avg us':>9s} {'TFLOP/s':>8s} {'ms/step':>8s}")
for (name, gf), (cnt, sec) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if sec / NS * 1e3 < 0.05:
        continue
    print(f"{name:28s} {gf:9.2f} {cnt / NS:13.1f} {sec / cnt * 1e6:9.1f} {gf / 1e3 / (sec / cnt) if sec else 0:8.1f} {sec / NS * 1e3:8.2f}")
