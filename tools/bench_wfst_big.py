#!/usr/bin/env python3
"""WFST search on a graph of the reference's vocabulary size: 125,078 words (the size of the
reference's words.txt) x synthetic word 3-gram (1 M bigrams + 1 M trigrams), compiled on this host by the native compiler
(tools/bench_graph_build.py: L o G, determinize-star, minimize-encoded, T o LG) and searched on one MI355X with the
production options: 32 utterances offline (clusters of 8 workgroups), and frame-by-frame streaming."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import b2t_native as N               # noqa: E402
import bench_graph_build as GB       # noqa: E402
import bench_wfst as BW              # noqa: E402
from wfst_decoder import WfstSearch  # noqa: E402


def decode(g, lp, lens, U, T, hash_size, cluster):
    lib = N.load()
    lib.b2t_wfst_set_cluster(cluster)
    try:
        S = WfstSearch(g, BW.Opt, U=U, max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24, hash_size=hash_size, prune_interval=25,
                       prune_min_fill=0.5)
        ts = []
        for rep in range(2):
            S.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            S.search(lp[:U].contiguous(), lens[:U]); torch.cuda.synchronize(); t1 = time.perf_counter()
            mem = S.memory_stats()
            hdr = S._header()
            if hdr[:, 3].any():
                return dict(error=f"overflow bits {int(np.bitwise_or.reduce(hdr[:, 3]))}", tokens=[m["created_tokens"] for m in mem][:4],
                            frames=S.frames_decoded()[:4], search_ms=round((t1 - t0) * 1e3, 1))
            fin = S.finalize(); t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1))
        frames = sum(S.frames_decoded())
        return dict(utterances=U, workgroups_per_utterance=int(lib.b2t_wfst_cluster_size(U)), hash_slots=S.caps[3],
                    search_ms=round(min(t[0] for t in ts) * 1e3, 2), finalize_nbest100_ms=round(min(t[1] for t in ts) * 1e3, 2),
                    search_ms_per_frame=round(min(t[0] for t in ts) * 1e3 / T, 3),
                    tokens_per_frame=round(sum(m["created_tokens"] for m in mem) / frames, 1),
                    links_per_frame=round(sum(m["created_links"] for m in mem) / frames, 1)), fin
    finally:
        lib.b2t_wfst_set_cluster(0)


def run(n_words=125078, n_grams=1000000, U=32, order=3):
    lib = N.load(); dev = torch.device("cuda:0")
    prons, words, arpa, g, st = GB.build(n_words, n_grams, optimize=True, order=order)
    _, _, _, _, seqs, logits, lens, _ = BW.make(U=U, seed=0, noise=0.9, graph=(prons, words, arpa, g), truth="lm", blank_boost=math.log(90.0))
    _, _, lp = BW._logp(logits, dev, lib)
    T = logits.shape[1]
    t0 = time.perf_counter(); g.to_device(dev); torch.cuda.synchronize(); up_s = time.perf_counter() - t0
    out = dict(build=st, upload_s=round(up_s, 2), hbm_graph_gb=round(g.nbytes() / 1e9, 3), frames_max=int(T))
    print(json.dumps(out), flush=True)
    for tag, u, hs, cl in (("u4_single", 4, 1 << 20, 1), ("u4_cluster", 4, 1 << 20, 0), ("u32_cluster", U, 1 << 18, 0), ("u32_single", U, 1 << 18, 1)):
        try:
            r = decode(g, lp, lens, u, T, hs, cl)
            if isinstance(r, tuple):
                r, fin = r
                wfst_1 = [[g.words[w] for w in f[0][2]] if f else [] for f in fin]
                r["wer_vs_truth"] = round(sum(BW.edit(h, q) for h, q in zip(wfst_1, seqs[:u])) / sum(len(q) for q in seqs[:u]), 4)
            out[tag] = r
        except Exception as e:       # noqa: BLE001
            out[tag] = dict(error=repr(e)[:300])
        print(tag, json.dumps(out[tag]), flush=True)
        torch.cuda.empty_cache()
    # round 4: the same graph with 10-byte arcs (b2t_wfst_graph_t.compact): graph bytes against search time
    try:
        import copy
        full_gb = g.nbytes() / 1e9
        g._dev = None; torch.cuda.empty_cache()
        gc = copy.copy(g); gc._dev = None; gc.set_compact(True)
        r = decode(gc, lp, lens, U, T, 1 << 18, 0)
        if isinstance(r, tuple):
            r, fin = r
            wfst_1 = [[g.words[w] for w in f[0][2]] if f else [] for f in fin]
            r["wer_vs_truth"] = round(sum(BW.edit(h, q) for h, q in zip(wfst_1, seqs[:U])) / sum(len(q) for q in seqs[:U]), 4)
        r["hbm_graph_gb"] = round(gc.nbytes() / 1e9, 3); r["hbm_graph_gb_full_width"] = round(full_gb, 3)
        out["u32_cluster_compact_arcs"] = r
        print("u32_cluster_compact_arcs", json.dumps(r), flush=True)
        gc._dev = None; torch.cuda.empty_cache()
    except Exception as e:           # noqa: BLE001
        out["u32_cluster_compact_arcs"] = dict(error=repr(e)[:300])
    try:
        Ss = WfstSearch(g, BW.Opt, U=U, max_frames=T + 8, max_tokens=1 << 21, max_links=1 << 23, hash_size=1 << 18, prune_interval=25)
        lat = []
        for t in range(T):
            fr = lp[:, t:t + 1].contiguous()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Ss.search(fr, np.minimum(1, np.maximum(0, lens - t)).astype(np.int32)); Ss.best_path(False, max_len=2 * T + 8)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat[5:]) * 1e3
        out["streaming_u32"] = dict(p50_ms_per_frame=round(float(np.percentile(lat, 50)), 3), p95_ms_per_frame=round(float(np.percentile(lat, 95)), 3))
    except Exception as e:           # noqa: BLE001
        out["streaming_u32"] = dict(error=repr(e)[:300])
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a), indent=1))
