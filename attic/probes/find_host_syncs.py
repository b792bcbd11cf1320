#!/usr/bin/env python3
"""Which Python lines of one bench training step issue blocking host<->device copies (they serialise host and GPU)?"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch, numpy as np
import bench, b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
dev = torch.device("cuda:0")
model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)
def step(i):
    feats = ops.augment_smooth(x, 2, 100, "same", cut=1, white_std=1.0, offset_std=0.2, seed=i)
    return ts.step(feats, days, labels, nts - 1, lens)
for i in range(3): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3); step(4)
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    n = e.name
    if "Memcpy" in n or "memcpy" in n or "hipMemcpy" in n or "hipStreamSynchronize" in n or "hipDeviceSynchronize" in n or "hipEventSynchronize" in n:
        key = (n, tuple(e.stack[:6]) if e.stack else ())
        seen[key] = seen.get(key, 0) + 1
for (n, st), c in sorted(seen.items(), key=lambda kv: -kv[1])[:20]:
    print(c, n)
    for s in st: print("      ", s)
