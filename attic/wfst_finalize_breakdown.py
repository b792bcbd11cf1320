import sys, os, time, math
sys.path.insert(0, "tools"); 
import bench_wfst as B
import torch, numpy as np
N, ops = B.N, B.ops
lib = N.load(); dev = torch.device("cuda:0"); _p = ops._p
prons, words, arpa, g, seqs, logits, lens, build_s = B.make()
U, T, C = logits.shape
lg = torch.from_numpy(logits).to(dev); pri = torch.zeros_like(lg); lp = torch.empty_like(lg)
N.check(lib.b2t_lm_prologue_f32(_p(lg), _p(pri), float(math.log(90.0)), _p(lp), U * T, C, ops._stream()), "prologue")
S = B.WfstSearch(g, B.Opt, U=U, max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24)
import ctypes as Cc
for rep in range(3):
    S.reset(); S.search(lp, lens); torch.cuda.synchronize(); t0 = time.perf_counter()
    N.check(lib.b2t_wfst_finalize(Cc.byref(S.cg), Cc.byref(S.co), ops._p(S.state), S.U, S._s()), "fin"); torch.cuda.synchronize(); t1 = time.perf_counter()
    S.finalized = True
    cn, host = S._lattices(); t2 = time.perf_counter()
    r = S._nbest_all(100); t3 = time.perf_counter()
    print("finalize kernel %.1f ms, lattice+copy %.1f ms, nbest_all (incl. lattices) %.1f ms; arcs max %d states max %d" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, cn[:,1].max(), cn[:,0].max()))
