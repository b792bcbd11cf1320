#!/usr/bin/env python3
"""Pins the n-gram arithmetic to the reference's SRILM (language_model/srilm-1.7.3, lm/src/NgramLM.cc:328-374
Ngram::wordProbBO): builds `ngram` / `ngram-count` from the sources under /root/reference with the committed recipe
oracle/Makefile.srilm (outputs in oracle/_ref/, git-ignored), trains ARPA models the way the reference's recipe does
(language_model/examples/speech/s0/local/build_lm.sh:36-46: ngram-count with -gt*min / -unk / -limit-vocab, then
ngram -prune), and records what `ngram -ppl -debug 2` prints for a set of test sentences: one log10 probability and the
n-gram order used for every word and for </s>.

Writes tests/golden/srilm_ngram.npz (data only: ARPA text SRILM wrote, sentences, per-word numbers).  Run in the build
container (needs /root/reference and g++); tests/test_srilm_golden.py checks ngram_lm.SparseNGramLM / NGramLM, the
oracle's ngram_log10 and wfst.grammar_fst against it on any machine.
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFBIN = os.path.join(ROOT, "oracle", "_ref")


def build():
    subprocess.check_call(["make", "-s", "-f", os.path.join(ROOT, "oracle", "Makefile.srilm"), "-j8"], cwd=ROOT)


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{cmd}: {r.stderr[-800:]}")
    return r.stdout + r.stderr


LINE = re.compile(r"p\( (\S+) \| .*\)\s+= \[(\w+)\] \S+ \[ (\S+) \]")


def ppl_debug2(order, arpa, sentences, tmp, unk=True):
    """-> per sentence: (words as SRILM saw them incl. </s>, log10 p, order used: 1..N, 0 = OOV)"""
    with open(os.path.join(tmp, "test.txt"), "w") as f:
        f.write("\n".join(" ".join(s) for s in sentences) + "\n")
    out = run([os.path.join(REFBIN, "ngram"), "-order", str(order), "-lm", arpa, "-ppl", "test.txt", "-debug", "2"] +
              (["-unk", "-map-unk", "<unk>"] if unk else []), tmp)
    res, cur = [], None
    for line in out.splitlines():
        m = LINE.search(line)
        if m:
            w, kind, lp = m.groups()
            k = 0 if kind == "OOV" else int(kind[0])
            cur.append((w, float(lp.replace("-inf", "-inf")), k))
        elif line and not line.startswith(("\t", " ", "file ", "reading ")) and "sentences," not in line and "zeroprobs" not in line:
            cur = []
            res.append(cur)
    res = [r for r in res if r]
    assert len(res) == len(sentences), (len(res), len(sentences), out[-2000:])
    return res


def markov_corpus(vocab, n_sent, rs, order_bias=0.75):
    """Sentences from a random sparse 2nd-order chain, so that bigrams / trigrams repeat and back-off happens."""
    V = len(vocab)
    fav = {(a, b): rs.choice(V, size=3, replace=False) for a in range(-1, V) for b in range(-1, V)}
    out = []
    for _ in range(n_sent):
        a, b, s = -1, -1, []
        for _ in range(int(rs.randint(3, 12))):
            w = int(rs.choice(fav[(a, b)])) if rs.rand() < order_bias else int(rs.randint(V))
            s.append(vocab[w]); a, b = b, w
        out.append(s)
    return out


def main():
    build()
    rs = np.random.RandomState(20260928)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        # ---- word level: the reference's recipe (build_lm.sh:36-46), order 3, vocabulary-limited with <unk>, then pruned
        words = [f"w{i:02d}" for i in range(36)] + ["the", "cat", "sat", "on"]
        corpus = markov_corpus(words + ["oovword"], 900, rs)
        with open(os.path.join(tmp, "corpus.txt"), "w") as f:
            f.write("\n".join(" ".join(s) for s in corpus) + "\n")
        with open(os.path.join(tmp, "lexicons.txt"), "w") as f:
            f.write("\n".join(words) + "\n")
        run([os.path.join(REFBIN, "ngram-count"), "-order", "3", "-gt1min", "0", "-gt2min", "1", "-gt3min", "1", "-unk", "-map-unk",
             "<unk>", "-limit-vocab", "-vocab", "lexicons.txt", "-text", "corpus.txt", "-lm", "lm_orig.arpa"], tmp)
        run([os.path.join(REFBIN, "ngram"), "-prune", "1e-4", "-order", "3", "-lm", "lm_orig.arpa", "-write-lm", "lm_pruned.arpa"], tmp)
        tests = markov_corpus(words, 40, rs, 0.6) + [["the", "cat", "sat", "on", "w03"], ["w01"], ["zzz", "w02", "qqq", "w05", "w06"],
                                                     ["w10"] * 6]
        for tag, arpa in (("w3", "lm_orig.arpa"), ("w3p", "lm_pruned.arpa")):
            res = ppl_debug2(3, arpa, tests, tmp)
            out[f"{tag}_arpa"] = np.array(open(os.path.join(tmp, arpa)).read())
            out[f"{tag}_words"] = np.array(words)
            out[f"{tag}_n"] = np.int64(len(tests))
            for i, (s, r) in enumerate(zip(tests, res)):
                assert len(r) == len(s) + 1
                out[f"{tag}_{i}_sent"] = np.array(s)
                out[f"{tag}_{i}_log10"] = np.array([x[1] for x in r], np.float64)
                out[f"{tag}_{i}_order"] = np.array([x[2] for x in r], np.int32)
        # ---- token level: order 5 over 40 phoneme-like tokens (what the prefix beam's NGramLM holds), Witten-Bell
        toks = [f"p{i:02d}" for i in range(1, 41)]
        corpus = markov_corpus(toks, 700, rs, 0.85)
        with open(os.path.join(tmp, "tok.txt"), "w") as f:
            f.write("\n".join(" ".join(s) for s in corpus) + "\n")
        run([os.path.join(REFBIN, "ngram-count"), "-order", "5", "-wbdiscount", "-text", "tok.txt", "-lm", "tok5.arpa"], tmp)
        tests = markov_corpus(toks, 30, rs, 0.7) + corpus[:15]       # training sentences too: 4- and 5-gram hits
        res = ppl_debug2(5, "tok5.arpa", tests, tmp, unk=False)
        out["t5_arpa"] = np.array(open(os.path.join(tmp, "tok5.arpa")).read())
        out["t5_words"] = np.array(toks)
        out["t5_n"] = np.int64(len(tests))
        for i, (s, r) in enumerate(zip(tests, res)):
            out[f"t5_{i}_sent"] = np.array(s)
            out[f"t5_{i}_log10"] = np.array([x[1] for x in r], np.float64)
            out[f"t5_{i}_order"] = np.array([x[2] for x in r], np.int32)
    path = os.path.join(HERE, "srilm_ngram.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
