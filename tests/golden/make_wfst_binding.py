#!/usr/bin/env python3
"""Fixture for the WFST search in the regime production runs in: the tools/bench_wfst.py graph
(400-word lexicon x word 3-gram, T o L o G 120 k states / 419 k arcs), production options (beam 17, max_active 7000,
min_active 200, lattice_beam 8, acoustic_scale 0.325: language-model-standalone.py:486-496), utterances noisy enough that
a frame holds more than max_active tokens, decoded by the ORACLE (oracle/wfst_oracle.py, the sequential restatement of
lattice-faster-decoder.cc; ~25 s of Python per utterance, hence a fixture).  Stored per utterance: the log-probabilities,
the oracle's 100-best list (words, graph / acoustic cost, alignment, times), and per decoded frame

  n_tok        tokens GetCutoff saw (lattice-faster-decoder.cc:650-720 counts every token of the hash list, including the
               over-the-cutoff ones ProcessEmitting :722-824 created before next_cutoff had tightened),
  n_extra      how many of them are such tokens (cost >= the frame's final next_cutoff),
  bound        1 if max_active decided the cutoff,
  differs      1 if GetCutoff's (cutoff, adaptive_beam) would differ with those tokens left out -- exactly the frames in
               which the data-parallel search (csrc/wfst.hip never creates them) can part from the sequential one.

Run here (CPU); tests/test_gpu_wfst.py::test_wfst_binding_regime_bench_graph compares the HIP search with it.
The oracle is parity-unpinned by the reference (no OpenFST in the image: the C++ cannot be built)."""
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "nejm-brain-to-text_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench_wfst as BW                  # noqa: E402
from oracle import wfst_oracle as W      # noqa: E402

U, NBEST = 5, 100
OPTS = dict(beam=17.0, max_active=7000, min_active=200, lattice_beam=8.0, acoustic_scale=0.325, nbest=NBEST, blank_skip_thresh=1.0)


class Instrumented(W.LatticeFasterDecoder):
    def init_decoding(self):
        self.extra_ids, self.stats = set(), []
        super().init_decoding()

    def get_cutoff(self, toks):
        res = super().get_cutoff(toks)
        real = [t for t in toks if id(t) not in self.extra_ids]
        res2 = super().get_cutoff(real) if len(real) != len(toks) else res
        costs = np.array([t.tot_cost for t in toks], np.float32)
        srt = np.sort(costs)
        bound = int(len(toks) > self.cfg.max_active and srt[self.cfg.max_active] < srt[0] + np.float32(self.cfg.beam))
        self.stats.append((len(toks), len(toks) - len(real), bound, int((float(res[0]), float(res[1])) != (float(res2[0]), float(res2[1])))))
        return res

    def process_emitting(self, loglike):
        nc = super().process_emitting(loglike)
        self.extra_ids = {id(t) for t in self.toks.values() if t.tot_cost >= nc}
        return nc


def main():
    prons, words, arpa, g, seqs, logits, lens, _ = BW.make(U=U, noise=1.0, seed=3)
    out = dict(n_utt=np.int64(U), opts=np.array([OPTS[k] for k in ("beam", "max_active", "min_active", "lattice_beam", "acoustic_scale")], np.float64),
               graph=np.array([g.n_states, g.n_arcs], np.int64), make_args=np.array([400, 3000, U, 3], np.int64), noise=np.float64(1.0))
    for u in range(U):
        lg = logits[u, :lens[u]]
        lp = (lg - np.log(np.exp(lg).sum(-1, keepdims=True))).astype(np.float32)
        lp[:, 0] -= np.float32(math.log(90.0))
        R = W.CtcWfstBeamSearch(g, W.Config(**OPTS))
        R.dec = Instrumented(g, R.cfg)
        R.reset()
        t0 = time.time()
        R.search(lp)
        part = (list(R.outputs[0]), R.likelihood[0])
        R.finalize_search()
        st = np.array(R.dec.stats, np.int64)
        print(f"utt {u}: {lp.shape[0]} frames, {time.time() - t0:.0f} s, tokens/frame max {st[:, 0].max()} mean {st[:, 0].mean():.0f}, "
              f"bound {st[:, 2].sum()}, extras {st[:, 1].sum()}, frames where extras change the cutoff {st[:, 3].sum()}, n-best {len(R.outputs)}")
        out[f"u{u}_logp"] = lp
        out[f"u{u}_truth"] = np.array(seqs[u])
        out[f"u{u}_stats"] = st
        out[f"u{u}_partial_words"] = np.array(part[0], np.int64)
        out[f"u{u}_partial_scores"] = np.array(part[1], np.float64)
        out[f"u{u}_n"] = np.int64(len(R.outputs))
        out[f"u{u}_scores"] = np.array(R.likelihood, np.float64)                     # (-graph, -acoustic)
        woff = np.cumsum([0] + [len(w) for w in R.outputs]); aoff = np.cumsum([0] + [len(a) for a in R.inputs])
        out[f"u{u}_woff"], out[f"u{u}_aoff"] = woff.astype(np.int64), aoff.astype(np.int64)
        out[f"u{u}_words"] = np.array([x for w in R.outputs for x in w], np.int64)
        out[f"u{u}_inputs"] = np.array([x for a in R.inputs for x in a], np.int64)
        out[f"u{u}_times"] = np.array([x for a in R.times for x in a], np.int64)
    path = os.path.join(HERE, "wfst_binding.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
