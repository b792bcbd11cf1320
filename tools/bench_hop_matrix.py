#!/usr/bin/env python3
"""Hop latency (event record on stream i -> wait on stream j -> tiny kernel) for all stream pairs of one process."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "16"))
import torch
dev = torch.device("cuda:0")
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
x = torch.zeros(1024, device=dev)
big_a = torch.randn(6144, 6144, device=dev); big_b = torch.randn(6144, 6144, device=dev)
streams = [torch.cuda.Stream() for _ in range(NS)]
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
N = 100
def run(ia, ib):
    best = 1e9
    for rep in range(2):
        with torch.cuda.stream(streams[ia]):
            for _ in range(3):
                big_a @ big_b
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        cur, other = streams[ia], streams[ib]
        for i in range(N):
            with torch.cuda.stream(cur):
                x.add_(1.0)
                if ia != ib:
                    ev = torch.cuda.Event(); ev.record()
            if ia != ib:
                other.wait_event(ev); cur, other = other, cur
        with torch.cuda.stream(cur):
            e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / N * 1e3)
    return best
print("us per (tiny kernel + hop); rows = stream i, cols = stream j")
for i in range(NS):
    print(f"{i:2d}: " + " ".join(f"{run(i, j):6.1f}" for j in range(NS)))
