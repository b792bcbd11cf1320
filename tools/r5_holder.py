#!/usr/bin/env python3
"""An idle process that HOLDS hardware queues on the GPU (round-5 slow-mode hunt): creates its HIP context, n streams with one tiny
kernel each (so the runtime maps them to hardware queues), then sleeps.  usage: r5_holder.py [n_streams=16] [seconds=120]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 120
x = torch.zeros(1024, device="cuda:0")
ss = [torch.cuda.Stream() for _ in range(n)]
for s in ss:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
print(f"holder: {n} streams live, GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}", flush=True)
time.sleep(sec)
