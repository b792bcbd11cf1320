#!/usr/bin/env python3
"""Timeline of the pipelined decode loop of tools/bench_wfst.py (search of batch b+1 under the host n-best of batch b): where a
batch's wall time goes on the main thread."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import b2t_native as N
import bench_wfst as BW
import wfst_decoder as WD
from wfst_decoder import WfstSearch


def main():
    lib = N.load(); dev = torch.device("cuda:0")
    prons, words, arpa, g, seqs, logits, lens, build_s = BW.make()
    U, T, Cc = logits.shape
    lg, pri, lp = BW._logp(logits, dev, lib)
    big = dict(max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24)
    S = [WfstSearch(g, BW.Opt, U=U, prune_interval=25, prune_min_fill=0.5, **big) for _ in range(2)]
    marks = []
    orig_lat = WfstSearch._lattices; orig_chk = WfstSearch._check_overflow; orig_nb = WfstSearch._nbest_host

    def lat(self, *a, **k):
        marks.append(("lat0", time.perf_counter())); r = orig_lat(self, *a, **k); marks.append(("lat1", time.perf_counter())); return r

    def chk(self, *a, **k):
        marks.append(("chk0", time.perf_counter())); r = orig_chk(self, *a, **k); marks.append(("chk1", time.perf_counter())); return r

    def nb(self, *a, **k):
        marks.append(("nb0", time.perf_counter())); r = orig_nb(self, *a, **k); marks.append(("nb1", time.perf_counter())); return r

    WfstSearch._lattices = lat; WfstSearch._check_overflow = chk; WfstSearch._nbest_host = nb
    import gc
    gcl = []
    def cb(phase, info):
        if phase == "start": cb.t = time.perf_counter()
        else: gcl.append((info["generation"], round((time.perf_counter() - cb.t) * 1e3, 2), info["collected"]))
    gc.callbacks.append(cb)
    pend = None
    for b in range(12):
        Sx = S[b % 2]
        marks.append((f"batch{b}", time.perf_counter()))
        Sx.reset(); Sx.search(lp, lens)
        marks.append(("searched", time.perf_counter()))
        f = Sx.finalize_async()
        marks.append(("fin_async_ret", time.perf_counter()))
        if pend is not None:
            pend.result()
        marks.append(("pend_done", time.perf_counter()))
        pend = f
    pend.result()
    bt = [t for n, t in marks if n.startswith("batch")]
    print("batch durations ms:", [round((y - x) * 1e3, 1) for x, y in zip(bt[:-1], bt[1:])])
    print("gc events (gen, ms, collected) over 1 ms:", [g for g in gcl if g[1] > 1.0], "n events", len(gcl), "tracked objects", len(gc.get_objects()))
    t0 = [t for n, t in marks if n == "batch3"][0]
    for n, t in marks:
        if t >= t0 and t < t0 + 0.13:
            print(f"{(t - t0) * 1e3:8.2f}  {n}")


if __name__ == "__main__":
    main()
