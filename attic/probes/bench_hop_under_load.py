#!/usr/bin/env python3
"""Cross-stream event-hop latency WHILE another stream runs a kernel with more workgroups than the chip has slots.
Hypothesis under test: a dispatch whose workgroups are still being launched occupies its hardware pipe, and queues that
share the pipe cannot start their next dispatch until it has finished launching (hop = O(duration of the big kernel))."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
dev = torch.device("cuda:0")
small_a = torch.randn(256, 256, device=dev); small_b = torch.randn(256, 256, device=dev)
big_n = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
big_a = torch.randn(big_n, big_n, device=dev); big_b = torch.randn(big_n, big_n, device=dev)
NS = 11
streams = [torch.cuda.Stream() for _ in range(NS)]
N = 60

def chain(sA, sB, n=N):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sA):
        e0.record()
    cur, other = sA, sB
    for i in range(n):
        with torch.cuda.stream(cur):
            small_a @ small_b
            ev = torch.cuda.Event(); ev.record()
        other.wait_event(ev); cur, other = other, cur
    with torch.cuda.stream(cur):
        e1.record()
    return e0, e1

for load in (False, True):
    for (ia, ib) in ((1, 2), (1, 3), (1, 4), (1, 5), (2, 6), (3, 7), (5, 9), (9, 10)):
        for il in ((0,) if load else (None,)):
            torch.cuda.synchronize()
            # a blocker in front of everything: the host enqueues the whole experiment while it runs
            with torch.cuda.stream(streams[ia]):
                for _ in range(6):
                    big_a @ big_b
                blk = torch.cuda.Event(); blk.record()
            if load:
                streams[il].wait_event(blk)
                with torch.cuda.stream(streams[il]):
                    l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    l0.record()
                    for _ in range(4):
                        big_a @ big_b
                    l1.record()
            e0, e1 = chain(streams[ia], streams[ib])
            torch.cuda.synchronize()
            msg = f"load {'on stream 0' if load else 'none':12s} hop pair ({ia},{ib}): {e0.elapsed_time(e1) / N * 1e3:8.1f} us per kernel+hop"
            if load:
                msg += f"   (load ran {l0.elapsed_time(l1):.2f} ms, chain {e0.elapsed_time(e1):.2f} ms)"
            print(msg)
