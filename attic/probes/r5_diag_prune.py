import os, sys, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/nejm-brain-to-text_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_wfst as TW
import ngram_lm, wfst, b2t_native as N, b2t_ops as ops
from oracle import wfst_oracle as W
from wfst_decoder import WfstSearch
lib = N.load()
prons = ngram_lm.synthetic_lexicon(60, 41, seed=11); words = sorted(prons)
arpa = ngram_lm.synthetic_word_arpa(words, 3, 400, seed=12)
g = wfst.build_tlg(prons, arpa, sil_prob=0.5)
rs = np.random.RandomState(91)
seqs, lps, batch, lens = TW.utterances(prons, words, 8, rs, noise=1.0, n_words=(1, 6))
lens = lens.copy(); lens[6] = 3
dev_batch = torch.from_numpy(batch).cuda(); T = batch.shape[1]
o = TW.Opt(nbest=30)
d = float(o.lattice_beam) * 0.1
cuts = [0, 11, 30, 31, 55, T]
for variant in ("single prune, no restore", "restore + second prune (mode 0 then 1)", "restore + second prune (mode 1 then 0)", "clone only"):
    os.environ["B2T_WFST_PRUNE_CLUSTER"] = "1"
    S = WfstSearch(g, o, U=8, max_frames=T + 8, prune_interval=0)
    for a, b in zip(cuts[:-1], cuts[1:]):
        S.search(dev_batch[:, a:b].contiguous(), np.clip(lens - a, 0, b - a))
        torch.cuda.synchronize()
        if variant.startswith("restore"):
            snap = S.state.clone()
            for mode in (("0", "1") if "0 then 1" in variant else ("1", "0")):
                os.environ["B2T_WFST_PRUNE_CLUSTER"] = mode
                S.state.copy_(snap)
                N.check(lib.b2t_wfst_prune(C.byref(S.cg), C.byref(S.co), ops._p(S.state), 8, C.c_float(d), C.c_float(0.0), S._s()), "prune")
                torch.cuda.synchronize()
        else:
            if variant == "clone only":
                snap = S.state.clone(); S.state.copy_(snap)
            N.check(lib.b2t_wfst_prune(C.byref(S.cg), C.byref(S.co), ops._p(S.state), 8, C.c_float(d), C.c_float(0.0), S._s()), "prune")
    fin = S.finalize()
    bad = []
    for u in range(8):
        R = W.CtcWfstBeamSearch(g, TW.cfg_of(o)); R.search(lps[u][:lens[u]]); R.finalize_search()
        try:
            TW.compare_lists(fin[u], R, f"u{u}")
        except AssertionError as e:
            bad.append((u, str(e)[:60]))
    print("cuts variant:", variant, "mismatches vs oracle:", bad, flush=True)
