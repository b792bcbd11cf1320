#!/usr/bin/env python3
"""Stress the training step: N steps of the bench workload (fp32 or B2T_AMP=1), status checked every 500 steps; prints the
slowest 100-step window (a hand-off stall shows as a window seconds long) and fails on a refused step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import bench, b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dev = torch.device("cuda:0")
torch.manual_seed(10)
if os.environ.get("B2T_AMP"):
    ops.set_amp(True)
if os.environ.get("B2T_STRESS_SHAPE") == "c3":     # the shipped shape: H = 768, patch 14 / 4, dropout 0.4 / 0.2
    model = GRUDecoder(bench.F, 768, bench.D, bench.C, 0.4, 0.2, bench.L, 14, 4).to(dev).train()
else:
    model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)
worst, t_all = 0.0, time.perf_counter()
for blk in range(N // 100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100):
        ts.step(ops.augment_smooth(x, 2, 100, "same", cut=i % 3, white_std=1.0, offset_std=0.2, seed=blk * 100 + i), days, labels, nts - i % 3, lens)
    torch.cuda.synchronize()
    worst = max(worst, (time.perf_counter() - t0) * 10)
    if blk % 5 == 4:
        ts.check_status(); model._ws.check_sync()
ts.check_status(); model._ws.check_sync()
print(f"{N} steps clean; mean {(time.perf_counter() - t_all) / N * 1e3:.2f} ms per step, slowest 100-step window {worst:.2f} ms per step")
