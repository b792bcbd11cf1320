#!/usr/bin/env python3
"""Streaming WFST decode only (tools/bench_wfst.py's last section): 32 utterances, one frame per call, partial best path read back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import b2t_native as N
import bench_wfst as BW
from wfst_decoder import WfstSearch
lib = N.load(); dev = torch.device("cuda:0")
prons, words, arpa, g, seqs, logits, lens, build_s = BW.make()
U, T, C = logits.shape
lg, pri, lp = BW._logp(logits, dev, lib)
for rep in range(2):
    Ss = WfstSearch(g, BW.Opt, U=U, prune_interval=25, prune_min_fill=0.0, max_frames=T + 8, max_tokens=1 << 20, max_links=1 << 22)
    lat, ls, lb = [], [], []
    for t in range(T):
        fr = lp[:, t:t + 1].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Ss.search(fr, np.minimum(1, np.maximum(0, lens - t)).astype(np.int32))
        t1 = time.perf_counter()
        bp = Ss.best_path(False, max_len=2 * T + 8)
        t2 = time.perf_counter()
        lat.append(t2 - t0); ls.append(t1 - t0); lb.append(t2 - t1)
    lat, ls, lb = (np.array(v[5:]) * 1e3 for v in (lat, ls, lb))
    print(f"streaming: p50 {np.percentile(lat, 50):.3f} ms per frame (p95 {np.percentile(lat, 95):.3f}, max {lat.max():.2f}); search enqueue p50 "
          f"{np.percentile(ls, 50):.3f}, best_path incl. wait p50 {np.percentile(lb, 50):.3f}; last words {bp[0][2][-3:]}")
    del Ss
