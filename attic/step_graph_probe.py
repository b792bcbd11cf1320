#!/usr/bin/env python3
"""Can the whole training step (augmentation, forward on four queues, CTC, backward, clip, AdamW) be captured into one hipGraph, and
is replaying it any faster than issuing it?  Measurement only: the replay freezes the step's scalars (LR, seeds)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return ts.step(ops.augment_smooth(x, 2, 100, "same", cut=1, white_std=1.0, offset_std=0.2, seed=i), days, labels, nts - 1, lens)
for i in range(8): step(i)
torch.cuda.synchronize()
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"eager: {timeit(lambda: step(3)):.3f} ms per step", flush=True)
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3): step(i)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        step(3)
    torch.cuda.synchronize()
    print("captured", flush=True)
    print(f"replay: {timeit(lambda: g.replay()):.3f} ms per step", flush=True)
    print(f"eager again: {timeit(lambda: step(3)):.3f} ms per step", flush=True)
    ts.check_status(); model._ws.check_sync()
    print("status clean")
except Exception as e:      # noqa: BLE001
    print("capture / replay failed:", repr(e)[:400])
