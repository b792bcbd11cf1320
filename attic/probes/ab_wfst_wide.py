#!/usr/bin/env python3
"""Wide WFST search (U utterances, one workgroup each): time of the library / env in effect."""
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
UW = int(os.environ.get("UW", "256")); rep = (UW + U - 1) // U
lpw = lp.repeat(rep, 1, 1)[:UW].contiguous(); lensw = np.tile(lens, rep)[:UW]
S = WfstSearch(g, BW.Opt, U=UW, max_frames=T + 8, max_tokens=1 << 21, max_links=1 << 23, prune_interval=0)
ts = []
for r in range(3):
    S.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    S.search(lpw, lensw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
bp = S.best_path(False, max_len=2 * T + 8)
print(f"U={UW} G={lib.b2t_wfst_cluster_size(UW)} g1={os.environ.get('B2T_WFST_G1')}: search min {min(ts) * 1e3:.2f} ms; u0 {str(bp[0])[:90]}", flush=True)
