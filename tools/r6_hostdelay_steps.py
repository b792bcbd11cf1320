#!/usr/bin/env python3
"""Round 6 (verdict item 1d): does a slow host cost GPU time in EVERY step or only while the host has not yet run ahead?  The headline
step, 60 steps: a HIP event behind every step (GPU time of step i = elapsed(e[i-1], e[i])), host time stamps where each step's
enqueue starts and ends, and -- at the start of step i -- how many earlier steps the GPU has not finished yet (how far ahead the
host runs).  Run under B2T_EXEC_HOST_DELAY_US=0 / 5 / 20."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
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
for i in range(5): step(i)
torch.cuda.synchronize()
N = int(os.environ.get("R6_STEPS", "60"))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
evs[0].record()
t_host, ahead = [], []
t0 = time.perf_counter()
for i in range(N):
    a = time.perf_counter()
    ahead.append(sum(1 for k in range(max(0, i - 8), i) if not evs[k + 1].query()))
    step(5 + i)
    evs[i + 1].record()
    t_host.append((a - t0, time.perf_counter() - t0))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ts.check_status()
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
print("R6DELAY " + json.dumps(dict(delay_us=int(os.environ.get("B2T_EXEC_HOST_DELAY_US", "0")), steps=N, ms_per_step=round(dt / N * 1e3, 3),
      host_enqueue_ms=round(float(np.mean([b - a for a, b in t_host])) * 1e3, 3),
      gpu_ms_first8=[round(v, 2) for v in gpu[:8]], gpu_ms_median_rest=round(float(np.median(gpu[8:])), 3),
      gpu_ms_p10_p90_rest=[round(float(np.percentile(gpu[8:], 10)), 3), round(float(np.percentile(gpu[8:], 90)), 3)],
      steps_host_is_ahead_at_enqueue=ahead[:12] + ["..."] + ahead[-4:])), flush=True)
