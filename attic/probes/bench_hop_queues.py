#!/usr/bin/env python3
"""Cross-stream event-hop latency as a function of how many HIP streams (hardware queues) the process has touched.
usage: bench_hop_queues.py NS [pair_a pair_b]   (one NS per process: queues cannot be destroyed reliably)"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "16"))
import torch
dev = torch.device("cuda:0")
NS = int(sys.argv[1]); ia = int(sys.argv[2]) if len(sys.argv) > 2 else 0; ib = int(sys.argv[3]) if len(sys.argv) > 3 else 1
small_a = torch.randn(256, 256, device=dev); small_b = torch.randn(256, 256, device=dev)
big_a = torch.randn(6144, 6144, device=dev); big_b = torch.randn(6144, 6144, device=dev)
streams = [torch.cuda.Stream() for _ in range(NS)]
for s in streams:
    with torch.cuda.stream(s):
        small_a @ small_b
torch.cuda.synchronize()
N = 100
res = []
for rep in range(3):
    with torch.cuda.stream(streams[ia]):
        for _ in range(6):
            big_a @ big_b
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    cur, other = streams[ia], streams[ib]
    for i in range(N):
        with torch.cuda.stream(cur):
            small_a @ small_b
            ev = torch.cuda.Event(); ev.record()
        other.wait_event(ev); cur, other = other, cur
    with torch.cuda.stream(cur):
        e1.record()
    # same thing on ONE stream for the kernel's own time
    with torch.cuda.stream(streams[ia]):
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for i in range(N):
            small_a @ small_b
        f1.record()
    torch.cuda.synchronize()
    res.append((e0.elapsed_time(e1) / N * 1e3, f0.elapsed_time(f1) / N * 1e3))
print(f"streams touched {NS:2d}, pair ({ia},{ib}): hop chain {min(r[0] for r in res):7.1f} us per kernel, same stream {min(r[1] for r in res):6.1f} us per kernel")
