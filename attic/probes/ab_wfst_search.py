#!/usr/bin/env python3
"""Offline WFST search time (32 utterances, production options) of the library B2T_LIB points at (or the in-tree one)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import b2t_native as N
import bench_wfst as BW
from wfst_decoder import WfstSearch
lib = N.load(); dev = torch.device("cuda:0")
prons, words, arpa, g, seqs, logits, lens, build_s = BW.make()
_, _, lp = BW._logp(logits, dev, lib)
T = logits.shape[1]; U = lp.shape[0]
for fill in (0.5, 0.0):
    S = WfstSearch(g, BW.Opt, U=U, prune_interval=25, prune_min_fill=fill, max_frames=T + 8, max_tokens=1 << 21, max_links=1 << 23)
    ts = []
    for rep in range(int(os.environ.get("REPS", "6"))):
        S.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        S.search(lp, lens); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    fin = S.finalize()
    print(f"prune_min_fill {fill}: search min {min(ts) * 1e3:.2f} ms  median {np.median(ts) * 1e3:.2f} ms; first {str(fin[0][0])[:100]}", flush=True)
    del S
