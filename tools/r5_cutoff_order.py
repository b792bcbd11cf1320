#!/usr/bin/env python3
"""Round-5 verdict 1d (CPU only): how much does the reference's ORDER-DEPENDENT next_cutoff tightening (lattice-faster-decoder.cc:786-810,
walked in HashList order, kaldi/util/hash-list-inl.h) change results once max_active binds?  The oracle runs both rules on the fuzz
graphs of tests/test_gpu_wfst.py -- "sequential" (the reference, hash-list order restated) and "final" (csrc/wfst.hip's data-parallel
rule: every candidate against the frame's final cutoff) -- and reports, per utterance: list lengths, first differing rank, whether the
best hypothesis and its cost agree.   usage: r5_cutoff_order.py [max_active ...]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ngram_lm, wfst
from oracle import wfst_oracle as W


def utterances(prons, words, U, rs, noise, blank_bias, n_words):
    seqs, lps = [], []
    for u in range(U):
        seq = [words[i] for i in rs.randint(len(words), size=rs.randint(*n_words))]
        frames = []
        for w in seq:
            for c in list(prons[w][0]) + [1]:
                frames += [c] * rs.randint(1, 3) + [0] * rs.randint(0, 3)
        lg = np.full((len(frames), 41), -2.0, np.float32)
        for t, c in enumerate(frames):
            lg[t, c] = 3.0
        lg += rs.standard_normal(lg.shape).astype(np.float32) * noise
        lp = lg - np.log(np.exp(lg).sum(-1, keepdims=True))
        lp[:, 0] -= blank_bias
        seqs.append(seq); lps.append(lp.astype(np.float32))
    return seqs, lps


def lists(g, cfg, lp):
    R = W.CtcWfstBeamSearch(g, cfg)
    R.search(lp); R.finalize_search()
    return [(tuple(o), -(l[0] + l[1])) for o, l in zip(R.outputs, R.likelihood)]


def compare(a, b, tol=2e-3):
    """first rank at which the two n-best lists differ (words, or cost by more than tol); len if one is a prefix of the other"""
    for k, ((wa, ca), (wb, cb)) in enumerate(zip(a, b)):
        if wa != wb or abs(ca - cb) > tol:
            return k
    return min(len(a), len(b)) if len(a) != len(b) else -1


def run(max_actives, seeds=(101, 202, 303), verbose=True):
    rows = []
    for ma in max_actives:
        for seed in seeds:
            rs = np.random.RandomState(seed)
            for case in range(3):
                n_words = int(rs.randint(20, 51)); order = int(rs.randint(2, 4))
                prons = ngram_lm.synthetic_lexicon(n_words, 41, seed=seed * 10 + case)
                words = sorted(prons)
                arpa = ngram_lm.synthetic_word_arpa(words, order, int(rs.randint(80, 300)), seed=seed * 10 + case + 1)
                g = wfst.build_tlg(prons, arpa, sil_prob=float(rs.choice([0.3, 0.5, 0.7])))
                U = int(rs.randint(1, 5))
                seqs, lps = utterances(prons, words, U, rs, float(rs.choice([0.4, 0.9, 1.4])), float(rs.choice([0.0, math.log(90.0)])), (1, 4))
                kw = dict(beam=float(rs.choice([8.0, 12.0, 17.0])), max_active=ma, min_active=int(rs.choice([0, 20, 200])),
                          lattice_beam=float(rs.choice([4.0, 8.0])), blank_skip_thresh=float(rs.choice([1.0, 0.98])),
                          length_penalty=float(rs.choice([0.0, -0.3])), nbest=int(rs.choice([5, 20])), acoustic_scale=0.325)
                if kw["min_active"] > ma:
                    kw["min_active"] = 0
                for u in range(U):
                    a = lists(g, W.Config(cutoff_rule="sequential", **kw), lps[u])
                    b = lists(g, W.Config(cutoff_rule="final", **kw), lps[u])
                    k = compare(a, b)
                    rows.append(dict(max_active=ma, seed=seed, case=case, utt=u, len_seq=len(a), len_final=len(b), first_diff=k,
                                     best_same=bool(a and b and a[0][0] == b[0][0] and abs(a[0][1] - b[0][1]) <= 2e-3)))
                    if verbose:
                        print(f"max_active {ma:5d} seed {seed} case {case} utt {u}: lists {len(a)} / {len(b)}, first differing rank "
                              f"{'none' if k < 0 else k}, best hypothesis {'same' if rows[-1]['best_same'] else 'DIFFERS'}", flush=True)
    return rows


if __name__ == "__main__":
    mas = [int(x) for x in sys.argv[1:]] or [60, 150, 400]
    rows = run(mas)
    for ma in mas:
        r = [x for x in rows if x["max_active"] == ma]
        nd = [x for x in r if x["first_diff"] >= 0]
        print(f"max_active {ma}: {len(r)} utterances, {len(nd)} with differing lists (first differing ranks {sorted(x['first_diff'] for x in nd)}), "
              f"best hypothesis differs in {sum(not x['best_same'] for x in r)}")
