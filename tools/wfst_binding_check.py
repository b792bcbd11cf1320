#!/usr/bin/env python3
"""HIP WFST search vs the oracle's fixture in the max_active-binding regime (tests/golden/wfst_binding.npz, made by
tests/golden/make_wfst_binding.py on the tools/bench_wfst.py graph): prints, per utterance, frames decoded, 1-best equality,
cost differences and n-best overlap.  The numbers quoted in DESIGN 7 come from here."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)
import bench_wfst as BW                  # noqa: E402
from wfst_decoder import WfstSearch      # noqa: E402


def load():
    Z = np.load(os.path.join(ROOT, "tests", "golden", "wfst_binding.npz"))
    nw, npo, U, seed = (int(v) for v in Z["make_args"])
    prons, words, arpa, g, seqs, logits, lens, _ = BW.make(n_words=nw, n_per_order=npo, U=U, seed=seed, noise=float(Z["noise"]))
    assert [g.n_states, g.n_arcs] == Z["graph"].tolist(), "the graph builder no longer reproduces the fixture's graph"
    return Z, g, U


def run(Z, g, U, dev="cuda:0"):
    class Opt:
        beam, max_active, min_active, lattice_beam, acoustic_scale = (float(Z["opts"][0]), int(Z["opts"][1]), int(Z["opts"][2]),
                                                                      float(Z["opts"][3]), float(Z["opts"][4]))
        ctc_blank_skip_threshold, length_penalty, nbest = 1.0, 0.0, 100
    lps = [Z[f"u{u}_logp"] for u in range(U)]
    T = max(l.shape[0] for l in lps)
    batch = np.zeros((U, T, 41), np.float32)
    lens = np.array([l.shape[0] for l in lps], np.int32)
    for u, l in enumerate(lps):
        batch[u, :l.shape[0]] = l
    S = WfstSearch(g, Opt, U=U, device=dev, max_frames=T + 8, max_tokens=1 << 21, max_links=1 << 23)
    S.search(torch.from_numpy(batch).to(dev), lens)
    part = S.best_path(False)
    fin = S.finalize()
    return S, part, fin


def report(Z, U, S, part, fin):
    rows = []
    for u in range(U):
        n = int(Z[f"u{u}_n"])
        woff = Z[f"u{u}_woff"]; W = Z[f"u{u}_words"]; sc = Z[f"u{u}_scores"]
        ref = [(tuple(W[woff[k]:woff[k + 1]].tolist()), -(sc[k, 0] + sc[k, 1])) for k in range(n)]
        got = [(tuple(e[2]), -(e[3] + e[4])) for e in fin[u]]
        st = Z[f"u{u}_stats"]
        ref_set = {w: c for w, c in ref}
        got_set = {w: c for w, c in got}
        common = [w for w, _ in ref if w in got_set]
        dc = max((abs(ref_set[w] - got_set[w]) for w in common), default=0.0)
        rows.append(dict(utt=u, frames=S.frames_decoded()[u], frames_ref=int(st.shape[0]), frames_bound=int(st[:, 2].sum()),
                         frames_order_dependent=int(st[:, 3].sum()), n_ref=n, n_got=len(got),
                         partial_same=list(part[u][2]) == Z[f"u{u}_partial_words"].tolist(),
                         best_same=bool(got and got[0][0] == ref[0][0]), best_cost_diff=abs(got[0][1] - ref[0][1]) if got else None,
                         top10_overlap=len({w for w, _ in ref[:10]} & {w for w, _ in got[:10]}) / 10.0,
                         top100_overlap=len(set(ref_set) & set(got_set)) / max(1, n),
                         max_cost_diff_common=dc,
                         same_order_prefix=next((k for k, (a, b) in enumerate(zip(ref, got)) if a[0] != b[0]), min(len(ref), len(got)))))
    return rows


if __name__ == "__main__":
    Z, g, U = load()
    S, part, fin = run(Z, g, U)
    for r in report(Z, U, S, part, fin):
        print(json.dumps(r))
