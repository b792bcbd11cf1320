#!/usr/bin/env python3
"""Hop latency between streams 1 and 2 (different pipes) while OTHER streams run back-to-back ~100 us kernels:
neighbours on the same pipes as 1 and 2 (streams 5, 6) or on the other two pipes (streams 3, 4 -> pipes 3, 0)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
ma = torch.randn(1024, 1024, device=dev); mb = torch.randn(1024, 1024, device=dev)      # ~30-100 us
big_a = torch.randn(6144, 6144, device=dev); big_b = torch.randn(6144, 6144, device=dev)
streams = [torch.cuda.Stream() for _ in range(12)]
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
N = 100
def run(ia, ib, busy, nbusy=400):
    best = 1e9
    for rep in range(2):
        with torch.cuda.stream(streams[ia]):
            for _ in range(4):
                big_a @ big_b
            blk = torch.cuda.Event(); blk.record()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for b in busy:
            streams[b].wait_event(blk)
            with torch.cuda.stream(streams[b]):
                for _ in range(nbusy):
                    ma @ mb
        cur, other = streams[ia], streams[ib]
        for i in range(N):
            with torch.cuda.stream(cur):
                x.add_(1.0)
                ev = torch.cuda.Event(); ev.record()
            other.wait_event(ev); cur, other = other, cur
        with torch.cuda.stream(cur):
            e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / N * 1e3)
    return best
print("hop 1<->2, no neighbours            : %.1f us" % run(1, 2, []))
print("hop 1<->2, busy streams 3,4 (other pipes): %.1f us" % run(1, 2, [3, 4]))
print("hop 1<->2, busy streams 5,6 (same pipes) : %.1f us" % run(1, 2, [5, 6]))
print("hop 1<->2, busy streams 5,6,9,10 (same pipes): %.1f us" % run(1, 2, [5, 6, 9, 10], 200))
print("hop 1<->2, busy streams 3,4,7,8 (other pipes): %.1f us" % run(1, 2, [3, 4, 7, 8], 200))
print("hop 1<->2, busy streams 3..11 : %.1f us" % run(1, 2, list(range(3, 12)), 100))
