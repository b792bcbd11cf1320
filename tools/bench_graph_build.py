#!/usr/bin/env python3
"""Graph compiler at the reference's vocabulary size: a 125 k-word lexicon (the size of
language_model/pretrained_language_models/openwebtext_1gram_lm_sil/words.txt; synthetic pronunciations and a synthetic
word 3-gram ARPA: the reference's LMs are not in the checkout) through make_tlg.sh's pipeline with the native compiler
(csrc/graphc.cpp).  Prints sizes and times of every stage; `--save path.npz` keeps the graph for tools/bench_wfst_big.py."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT):
    sys.path.insert(0, p)
import ngram_lm   # noqa: E402
import wfst       # noqa: E402


def build(n_words, n_per_order, optimize=True, seed=0, order=3):
    t0 = time.time()
    prons = ngram_lm.synthetic_lexicon(n_words, 41, seed=seed + 1)
    words = sorted(prons)
    arpa = ngram_lm.synthetic_word_arpa(words, order, n_per_order, seed=seed + 2)
    t1 = time.time()
    st = {}
    g = wfst.build_tlg_native(prons, arpa, optimize=optimize, stats=st)
    st["s_synthesize_lexicon_and_arpa"] = round(t1 - t0, 2)
    st["arpa_mb"] = round(len(arpa) / 1e6, 1)
    return prons, words, arpa, g, st


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--words", type=int, default=125078)
    ap.add_argument("--ngrams", type=int, default=1000000, help="bigrams and trigrams each")
    ap.add_argument("--order", type=int, default=3)
    ap.add_argument("--plain", action="store_true", help="skip determinize-star / minimize-encoded")
    ap.add_argument("--save", default="")
    a = ap.parse_args()
    prons, words, arpa, g, st = build(a.words, a.ngrams, optimize=not a.plain, order=a.order)
    print(json.dumps(st, indent=1))
    if a.save:
        wfst.save_graph(g, a.save)
