#!/usr/bin/env python3
"""Phase ticks of the cluster search (a -DB2T_WFST_TIMING build of csrc/wfst.hip selected by B2T_LIB): one offline search of the bench workload
without prune passes (one launch per 25 frames); the kernel prints member 0 / utterance 0's accumulated 100 MHz ticks per phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import b2t_native as N
import bench_wfst as BW
from wfst_decoder import WfstSearch
lib = N.load(); dev = torch.device("cuda:0")
prons, words, arpa, g, seqs, logits, lens, _ = BW.make()
U, T, C = logits.shape
_, _, lp = BW._logp(logits, dev, lib)
S = WfstSearch(g, BW.Opt, U=U, max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24, prune_interval=0)
for rep in range(2):
    S.reset(); S.search(lp, lens); torch.cuda.synchronize()
    print("---- rep", rep, flush=True)
