#!/usr/bin/env python3
"""Latency of the WFST search for FEW utterances (the real-time decoder's case) by cluster size: U = 1, 4, 8 with 8 / 16 / 32
workgroups per utterance; offline (one call) and streaming (one frame per call + partial best path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import b2t_native as N
import bench_wfst as BW
from wfst_decoder import WfstSearch
lib = N.load(); dev = torch.device("cuda:0")
prons, words, arpa, g, seqs, logits, lens, build_s = BW.make()
_, _, lp_all = BW._logp(logits, dev, lib)
T = logits.shape[1]
for U in (1, 4, 8):
    lp = lp_all[:U].contiguous(); ln = lens[:U]
    for G in (8, 16, 32):
        lib.b2t_wfst_set_cluster(G)
        S = WfstSearch(g, BW.Opt, U=U, prune_interval=25, prune_min_fill=0.5, max_frames=T + 8, max_tokens=1 << 21, max_links=1 << 23)
        ts = []
        for rep in range(4):
            S.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            S.search(lp, ln); torch.cuda.synchronize(); t1 = time.perf_counter()
            fin = S.finalize(); t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1))
        best = [w for w in fin[0][0][2]][:4]
        S.reset(); lat = []
        for t in range(T):
            fr = lp[:, t:t + 1].contiguous()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            S.search(fr, np.minimum(1, np.maximum(0, ln - t)).astype(np.int32)); S.best_path(False, max_len=2 * T + 8)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat[5:]) * 1e3
        print(f"U={U} G={G}: search {min(t[0] for t in ts) * 1e3:.2f} ms, finalize + n-best {min(t[1] for t in ts) * 1e3:.2f} ms; "
              f"streaming p50 {np.percentile(lat, 50):.3f} ms per frame (p95 {np.percentile(lat, 95):.3f}); words {best}", flush=True)
        del S
lib.b2t_wfst_set_cluster(0)
