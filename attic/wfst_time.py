import os, sys, time, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import b2t_native as N, b2t_ops as ops, bench_wfst as BW
from wfst_decoder import WfstSearch
lib = N.load()
prons, words, arpa, g, seqs, logits, lens, _ = BW.make()
U, T, C = logits.shape
lg = torch.from_numpy(logits).cuda(); pri = torch.zeros_like(lg); lp = torch.empty_like(lg)
N.check(lib.b2t_lm_prologue_f32(ops._p(lg), ops._p(pri), float(math.log(90.0)), ops._p(lp), U * T, C, ops._stream()), "prologue")
for G in (1, 2, 4, 8):
    for iv in (0, 25):
        lib.b2t_wfst_set_cluster(G)
        S = WfstSearch(g, BW.Opt, U=U, max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24, prune_interval=iv)
        ts = []
        for rep in range(3):
            S.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            S.search(lp, lens); torch.cuda.synchronize(); t1 = time.perf_counter()
            # prune alone
            S.prune(); torch.cuda.synchronize(); t2 = time.perf_counter()
            S.finalize(); torch.cuda.synchronize(); t3 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1, t3 - t2))
        print(f"G={G} prune_interval={iv}: search {min(t[0] for t in ts)*1e3:.1f} ms, one more prune {min(t[1] for t in ts)*1e3:.1f} ms, finalize+nbest {min(t[2] for t in ts)*1e3:.1f} ms", flush=True)
        del S
