import os, sys, time, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import b2t_native as N, bench_wfst as BW
from wfst_decoder import WfstSearch
lib = N.load(); dev = torch.device("cuda:0")
prons, words, arpa, g, seqs, logits, lens, _ = BW.make()
U, T, C = logits.shape
_, _, lp = BW._logp(logits, dev, lib)
big = dict(max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24)
S = [WfstSearch(g, BW.Opt, U=U, prune_interval=25, prune_min_fill=0.5, **big) for _ in range(2)]
for s in S:
    s.reset(); s.search(lp, lens); s.finalize()
# sequential
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in range(6):
    S[0].reset(); S[0].search(lp, lens); S[0].finalize()
seq = (time.perf_counter() - t0) / 6 * 1e3
pend = None; torch.cuda.synchronize(); t0 = time.perf_counter()
for b in range(6):
    Sx = S[b % 2]
    Sx.reset(); Sx.search(lp, lens)
    f = Sx.finalize_async()
    if pend is not None: pend.result()
    pend = f
pend.result()
pipe = (time.perf_counter() - t0) / 6 * 1e3
# phases of finalize_async
Sx = S[0]; Sx.reset(); Sx.search(lp, lens); torch.cuda.synchronize()
import ctypes as Ct
t0 = time.perf_counter()
N.check(Sx.lib.b2t_wfst_finalize(Ct.byref(Sx.cg), Ct.byref(Sx.co), Sx.state.data_ptr(), Sx.U, None), "f"); torch.cuda.synchronize(); t1 = time.perf_counter()
hdr = Sx._header(); t2 = time.perf_counter()
cn, host = Sx._lattices(); t3 = time.perf_counter()
print(f"threads {os.environ.get('B2T_HOST_THREADS')}: sequential {seq:.1f} ms/batch, pipelined {pipe:.1f} ms/batch; finalize kernel {1e3*(t1-t0):.2f}, header {1e3*(t2-t1):.2f}, lattices (kernel + copies) {1e3*(t3-t2):.2f} ms; arcs max {int(cn[:,1].max())} sum {int(cn[:,1].sum())}")
