#!/usr/bin/env python3
"""B2T_EXEC_GRAPH=1: the passes of the bench step as replayed hipGraphs -- loss trajectory against the eager plan (must be
identical), ms per step, host enqueue per step, graphs built / passes replayed.  usage: r4_graph_probe.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import bench, b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.manual_seed(10)
if os.environ.get("B2T_AMP"): ops.set_amp(True)
model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)
def step(i):
    return ts.step(ops.augment_smooth(x, 2, 100, "same", cut=i % 3, white_std=1.0, offset_std=0.2, seed=i), days, labels, nts - i % 3, lens)
losses = []
for i in range(8): losses.append(step(i)[0])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N): losses.append(step(8 + i)[0])
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ts.check_status()
print(f"graph={os.environ.get('B2T_EXEC_GRAPH', '0')}: {dt / N * 1e3:.3f} ms per step, host enqueue {t_enq / N * 1e3:.3f} ms per step, "
      f"graphs built / replays / failed {model._ws.graph_stats()}, losses {[round(float(l), 4) for l in losses[-4:]]} sum {sum(float(l) for l in losses):.6f}")
