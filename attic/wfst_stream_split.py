#!/usr/bin/env python3
"""Streaming WFST decode, one frame per call: where a frame's time goes (search kernel vs partial best path), by frame index."""
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
    ls, lb = [], []
    for t in range(T):
        fr = lp[:, t:t + 1].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Ss.search(fr, np.minimum(1, np.maximum(0, lens - t)).astype(np.int32)); torch.cuda.synchronize()
        t1 = time.perf_counter()
        bp = Ss.best_path(False, max_len=2 * T + 8)
        t2 = time.perf_counter()
        ls.append(t1 - t0); lb.append(t2 - t1)
    ls, lb = (np.array(v) * 1e3 for v in (ls, lb))
    sel = [i for i in range(5, T) if i % 25 != 0]
    print(f"search+sync p50 {np.percentile(ls[sel], 50):.3f} ms; best_path p50 {np.percentile(lb[sel], 50):.3f} ms; best_path by frame: "
          + " ".join(f"{i}:{lb[i]:.3f}" for i in range(10, T, 20)))
    del Ss
