#!/usr/bin/env python3
"""Host-side profile of Rescore() (b2t_lattice_rescore_nbest_host) on the 32 lattices attic/dump_lattice.py saved
(gpurun_out/lattices.npz): time per utterance, product size, and the n-best lists for comparing two builds (B2T_LIB)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, ROOT)
import bench_wfst as B
import b2t_native as N, ngram_lm, wfst
lib = N.load()
Z = np.load(os.path.join(ROOT, "gpurun_out", "lattices.npz"))
prons, words, arpa, g, *_ = B.make(U=1)
word_id = {w: i for i, w in enumerate(g.words) if 0 < i <= len(words)}
wd0 = g.words.index("#0")
G_old = wfst.HostFst.from_fst(wfst.grammar_fst(arpa, word_id, wd0)).arcsort()
G_new = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(words, 4, 6000, seed=77), word_id, wd0)).arcsort()
nbest, beam = 100, 8.0
P = lambda a: a.ctypes.data_as(C.c_void_p)
tot, out = 0.0, []
U = len([k for k in Z.files if k.endswith("_meta")])
for u in range(U):
    a = {k: np.ascontiguousarray(Z[f"u{u}_{k}"]) for k in ("src", "dst", "il", "ol", "gr", "ac", "fs", "fc", "meta")}
    n_states, n_arcs, n_final, start, F = (int(v) for v in a["meta"])
    cap = nbest * (2 * F + 16) + 16
    ow = np.empty(cap, np.int32); oa = np.empty(cap, np.int32); woff = np.zeros(nbest + 1, np.int32); aoff = np.zeros(nbest + 1, np.int32)
    costs = np.empty(2 * nbest, np.float32); st = (C.c_longlong * 4)()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        n = lib.b2t_lattice_rescore_nbest_host(n_states, start, n_arcs, P(a["src"]), P(a["dst"]), P(a["il"]), P(a["ol"]), P(a["gr"]), P(a["ac"]), n_final,
                                               P(a["fs"]), P(a["fc"]), G_old._h, G_new._h, wd0, nbest, C.c_float(beam), P(ow), P(woff), cap, P(oa), P(aoff), cap,
                                               P(costs), st)
        best = min(best, time.perf_counter() - t0)
    assert n >= 0, N.last_error()
    tot += best
    out.append([(tuple(ow[woff[i]:woff[i + 1]].tolist()), round(float(costs[2 * i]), 3), round(float(costs[2 * i + 1]), 3), int(aoff[i + 1] - aoff[i])) for i in range(n)])
    if u < 4 or os.environ.get("B2T_VERBOSE"):
        print(f"u{u}: lattice {n_states} states / {n_arcs} arcs -> product {st[0]} / {st[1]}; {n} hypotheses; {best * 1e3:.2f} ms")
print(f"rescore 100-best: {tot * 1e3:.1f} ms for {U} utterances ({tot / U * 1e3:.2f} ms each, serial)")
if len(sys.argv) > 1:
    import pickle
    if os.path.exists(sys.argv[1]):
        ref = pickle.load(open(sys.argv[1], "rb"))
        same = sum(1 for a, b in zip(ref, out) if [x[0] for x in a] == [x[0] for x in b])
        cost_ok = all(abs(x[1] - y[1]) < 2e-3 and abs(x[2] - y[2]) < 2e-3 for a, b in zip(ref, out) for x, y in zip(a, b) if x[0] == y[0])
        print(f"vs {sys.argv[1]}: {same}/{U} utterances with identical word-sequence lists; costs of matching entries equal: {cost_ok}")
        for a, b in zip(ref, out):
            if [x[0] for x in a] != [x[0] for x in b]:
                k = next(i for i, (x, y) in enumerate(zip(a, b)) if x[0] != y[0]) if any(x[0] != y[0] for x, y in zip(a, b)) else min(len(a), len(b))
                print("  first difference at rank", k, "lens", len(a), len(b), a[k][:3] if k < len(a) else None, b[k][:3] if k < len(b) else None)
                break
    else:
        pickle.dump(out, open(sys.argv[1], "wb")); print("saved", sys.argv[1])
