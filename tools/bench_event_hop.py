#!/usr/bin/env python3
"""Latency of a cross-stream dependency (event record on stream A -> stream B waits -> kernel on B) vs the same two
kernels back to back on one stream.  Measured with device-side events around a chain of N hops."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
dev = torch.device("cuda:0")
a = torch.randn(1024, 1024, device=dev); b = torch.randn(1024, 1024, device=dev)
def work(): return a @ b       # ~15 us kernel
N = 200
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for mode in ("same stream", "ping-pong between two streams"):
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sA):
            e0.record()
        cur, other = sA, sB
        for i in range(N):
            with torch.cuda.stream(cur):
                work()
                if mode != "same stream":
                    ev = torch.cuda.Event(); ev.record()
            if mode != "same stream":
                other.wait_event(ev); cur, other = other, cur
        with torch.cuda.stream(cur):
            e1.record()
        torch.cuda.synchronize()
    print(f"{mode:32s}: {e0.elapsed_time(e1) / N * 1e3:7.1f} us per kernel")
