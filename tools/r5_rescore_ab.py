"""Rescore() of ONE dumped bench lattice (U=index; attic/dump_lattice.py -> gpurun_out/lattices.npz) through several builds of the
library loaded into one process and called alternately -- the build VM's speed drifts, an A/B in two processes does not hold.
usage: [U=0] [REPS=20] [B2T_RESCORE_THREADS=n] python tools/r5_rescore_ab.py lib_a.so lib_b.so ...   (see tools/tsan_rescore.sh)"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, ROOT)
import bench_wfst as B
import b2t_native as N, ngram_lm, wfst
lib0 = N.load()
libs = [C.CDLL(p) for p in sys.argv[1:]]
Z = np.load(os.environ.get("B2T_LATTICES", os.path.join(ROOT, "gpurun_out", "lattices.npz")))
prons, words, arpa, g, *_ = B.make(U=1)
word_id = {w: i for i, w in enumerate(g.words) if 0 < i <= len(words)}
wd0 = g.words.index("#0")
G_old = wfst.HostFst.from_fst(wfst.grammar_fst(arpa, word_id, wd0)).arcsort()
G_new = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(words, 4, 6000, seed=77), word_id, wd0)).arcsort()
nbest, beam = 100, 8.0
P = lambda a: a.ctypes.data_as(C.c_void_p)
u = int(os.environ.get("U", "0"))
a = {k: np.ascontiguousarray(Z[f"u{u}_{k}"]) for k in ("src", "dst", "il", "ol", "gr", "ac", "fs", "fc", "meta")}
n_states, n_arcs, n_final, start, F = (int(v) for v in a["meta"])
cap = nbest * (2 * F + 16) + 16
ow = np.empty(cap, np.int32); oa = np.empty(cap, np.int32); woff = np.zeros(nbest + 1, np.int32); aoff = np.zeros(nbest + 1, np.int32)
costs = np.empty(2 * nbest, np.float32); st = (C.c_longlong * 4)()
ts = [[] for _ in libs]
for rep in range(int(os.environ.get("REPS", "20"))):
    for k, lib in enumerate(libs):
        f = lib.b2t_lattice_rescore_nbest_host
        t0 = time.perf_counter()
        n = f(n_states, start, n_arcs, P(a["src"]), P(a["dst"]), P(a["il"]), P(a["ol"]), P(a["gr"]), P(a["ac"]), n_final,
              P(a["fs"]), P(a["fc"]), C.c_void_p(G_old._h), C.c_void_p(G_new._h), wd0, nbest, C.c_float(beam), P(ow), P(woff), cap, P(oa), P(aoff), cap, P(costs), st)
        ts[k].append((time.perf_counter() - t0) * 1e3)
for p, t in zip(sys.argv[1:], ts):
    t = sorted(t)
    print(f"{os.path.basename(p)}: min {t[0]:.1f}  q25 {t[len(t)//4]:.1f}  median {t[len(t)//2]:.1f} ms")
