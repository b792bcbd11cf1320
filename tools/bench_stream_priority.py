#!/usr/bin/env python3
"""Does HIP stream priority order workgroup dispatch between two kernels that both want the whole chip?
Two streams launch the same large matmul at the same moment (behind a common blocker); per-stream completion times."""
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
dev = torch.device("cuda:0")
n = 6144
a = torch.randn(n, n, device=dev); b = torch.randn(n, n, device=dev)
blk_a = torch.randn(8192, 8192, device=dev)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
def run(p1, p2):
    s0 = torch.cuda.Stream(); s1 = torch.cuda.Stream(priority=p1); s2 = torch.cuda.Stream(priority=p2)
    for s in (s1, s2):
        with torch.cuda.stream(s):
            a @ b
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        with torch.cuda.stream(s0):
            blk_a @ blk_a
            start = torch.cuda.Event(enable_timing=True); start.record()
        ends = []
        for s in (s1, s2):
            s.wait_event(start)
            with torch.cuda.stream(s):
                for _ in range(2):
                    a @ b
                e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e)
        torch.cuda.synchronize()
        res.append((start.elapsed_time(ends[0]), start.elapsed_time(ends[1])))
    return res[-1]
alone = None
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    a @ b; torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); a @ b; a @ b; e1.record()
torch.cuda.synchronize()
print("two matmuls alone: %.2f ms" % e0.elapsed_time(e1))
for p1, p2 in ((0, 0), (-1, 0), (0, -1), (-1, -1)):
    r = run(p1, p2)
    print(f"priorities ({p1:2d},{p2:2d}): stream 1 done after {r[0]:.2f} ms, stream 2 after {r[1]:.2f} ms")
