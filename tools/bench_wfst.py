#!/usr/bin/env python3
"""WFST decode on one MI355X (BASELINE configs[3]/[4] with the reference's searcher: token passing over T o L o G with the
production options, language-model-standalone.py:486-496) and the measured gap of the lexicon prefix beam that stood in
for it in round 1:
  * offline: 32 utterances in one call (search + finalize + n-best 100) -> ms per utterance
  * streaming: 32 concurrent utterances, one frame per call, partial best path read back every frame -> p50 / p95 per frame
  * agreement: 1-best word error rate and n-best overlap of b2t_prefix_beam_search_lex_f32 (beams 10/16 and 16/64) against
    the WFST search, and of the WFST search against the spelled truth
Synthetic lexicon + word 3-gram (the reference's LMs are not in the checkout); sizes are printed."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import b2t_native as N      # noqa: E402
import b2t_ops as ops       # noqa: E402
import ngram_lm             # noqa: E402
import wfst                 # noqa: E402
from wfst_decoder import WfstSearch, _pool_threads   # noqa: E402
import lm_decoder           # noqa: E402


def edit(a, b):
    prev = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        cur = [i] + [0] * len(b)
        for j in range(1, len(b) + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return prev[len(b)]


def sample_sentence(lm, rs, n):
    """n words drawn from the word n-gram itself (explicit continuations of the current history, proportional to their
    probability; back off when the history has none): sentences the LM likes, as real text is to a real LM."""
    s, out = lm.start_state, []
    for _ in range(n):
        st = s
        while True:
            a, e = int(lm.cb[st]), int(lm.ce[st])
            toks, nodes = lm.ctok[a:e], lm.cnode[a:e]
            keep = toks < lm.W
            if st == 0 or (keep.any() and rs.rand() < 0.85):
                break
            st = int(lm.suffix[st])
        pr = np.exp(lm.logp[nodes[keep]].astype(np.float64)); pr /= pr.sum()
        k = int(rs.choice(int(keep.sum()), p=pr))
        out.append(int(toks[keep][k])); s = int(lm.nstate[nodes[keep][k]])
    return out


def make(n_words=int(os.environ.get("B2T_WFST_WORDS", "400")), n_per_order=3000, U=32, seed=0, noise=0.9, graph=None, truth="uniform", blank_boost=0.0):
    t0 = time.time()
    if graph is not None:
        prons, words, arpa, g = graph
        n_words = len(words)
    else:
        prons = ngram_lm.synthetic_lexicon(n_words, 41, seed=seed + 1)
        words = sorted(prons)
        arpa = ngram_lm.synthetic_word_arpa(words, 3, n_per_order, seed=seed + 2)
        g = wfst.build_tlg(prons, arpa, sil_prob=0.5)
    build_s = time.time() - t0
    rs = np.random.RandomState(seed)
    seqs, rows = [], []
    wlm = ngram_lm.SparseNGramLM.from_arpa(arpa, words) if truth == "lm" else None
    for u in range(U):
        if wlm is not None:
            seq = [words[i] for i in sample_sentence(wlm, rs, int(rs.randint(4, 9)))]
        else:
            seq = [words[i] for i in rs.randint(n_words, size=rs.randint(4, 9))]
        frames = []
        for w in seq:
            prev = -1
            for c in list(prons[w][0]) + [1]:
                if c == prev:
                    frames.append(0)
                frames += [c] * rs.randint(1, 3) + [0] * rs.randint(0, 2)
                prev = c
        lg = np.full((len(frames), 41), -2.0, np.float32)
        for t, c in enumerate(frames):
            lg[t, c] = 3.0 + (blank_boost if c == 0 else 0.0)
        lg += rs.standard_normal(lg.shape).astype(np.float32) * noise
        seqs.append(seq); rows.append(lg)
    T = max(r.shape[0] for r in rows)
    logits = np.zeros((U, T, 41), np.float32); logits[:, :, 0] = 5.0      # padding frames: blank
    lens = np.array([r.shape[0] for r in rows], np.int32)
    for u, r in enumerate(rows):
        logits[u, :r.shape[0]] = r
    return prons, words, arpa, g, seqs, logits, lens, build_s


class Opt:
    max_active, min_active, beam, lattice_beam, acoustic_scale = 7000, 200, 17.0, 8.0, 0.325
    ctc_blank_skip_threshold, length_penalty, nbest = 1.0, 0.0, 100


def _logp(logits, dev, lib):
    lg = torch.from_numpy(logits).to(dev); pri = torch.zeros_like(lg); lp = torch.empty_like(lg)
    U, T, C = logits.shape
    N.check(lib.b2t_lm_prologue_f32(ops._p(lg), ops._p(pri), float(math.log(90.0)), ops._p(lp), U * T, C, ops._stream()), "prologue")
    return lg, pri, lp


def accuracy(prons, words, arpa, g, noise_levels=(0.9, 1.5, 2.0, 3.0), U=32, seed=0):
    """BASELINE configs[3] asks for WER / PER against the language_model/ reference.  With the reference's searcher rebuilt
    (WFST token passing) and its data absent, what can be reported is: word error rate AGAINST THE SPELLED TRUTH of the WFST
    1-best and of the lexicon prefix beam (b2t_prefix_beam_search_lex_f32: pronunciation trie + word 3-gram in HBM) at beams
    10/16 and 10/100, at three noise levels, plus how far the prefix beam's answers are from the graph search's."""
    lib = N.load(); dev = torch.device("cuda:0")
    C = 41
    lex = ngram_lm.Lexicon(prons, C)
    wlm = ngram_lm.SparseNGramLM.from_arpa(arpa, lex.words)
    rows = []
    for noise in noise_levels:
        # blank frames carry the margin real CTC outputs have (the decoder's blank penalty ln 90 is tuned for it): without
        # it every blank frame is a coin flip after the penalty and no searcher with a beam of 100 prefixes can cope
        _, _, _, _, seqs, logits, lens, _ = make(U=U, seed=seed, noise=noise, graph=(prons, words, arpa, g), truth="lm",
                                                 blank_boost=math.log(90.0))
        _, _, lp = _logp(logits, dev, lib)
        T = logits.shape[1]
        S = WfstSearch(g, Opt, U=U, max_frames=T + 8, max_tokens=1 << 21, max_links=1 << 23)
        S.search(lp, lens)
        fin = S.finalize()
        del S
        wfst_1 = [[g.words[w] for w in f[0][2]] if f else [] for f in fin]
        nref = sum(len(r) for r in seqs)
        row = dict(noise=noise, wfst_wer_vs_truth=round(sum(edit(h, r) for h, r in zip(wfst_1, seqs)) / nref, 4))
        for fb, sb in ((10, 16), (10, 100)):
            opts = lm_decoder.DecodeOptions(7000, 200, 17.0, 8.0, 0.325, 1.0, 0.0, 100)
            opts.first_beam_size, opts.second_beam_size = fb, sb
            opts.lm_alpha, opts.lm_beta, opts.lm_eos = 1.0 / 0.325, 0.0, True     # graph cost : acoustic cost = 1 : 0.325
            res = lm_decoder.DecodeResource("", "", "", "", "")
            res.set_lexicon_lm(lex, wlm, sil=1)
            e_truth = e_wfst = overlap = n_over = 0
            t0 = time.perf_counter()
            for u in range(U):
                dec = lm_decoder.BrainSpeechDecoder(res, opts, max_len=T + 8)
                dec.Decode(lp[u, :lens[u]])
                hyp = dec.result()
                h1 = hyp[0].sentence.split() if hyp else []
                e_truth += edit(h1, seqs[u]); e_wfst += edit(h1, wfst_1[u])
                ref_set = set(" ".join(g.words[w] for w in e[2]) for e in fin[u][:10])
                overlap += len(ref_set & set(r.sentence for r in hyp[:10])); n_over += len(ref_set)
            dt = time.perf_counter() - t0
            row[f"lexicon_prefix_beam_{fb}_{sb}"] = dict(wer_vs_truth=round(e_truth / nref, 4),
                                                         wer_vs_wfst_1best=round(e_wfst / max(1, sum(len(w) for w in wfst_1)), 4),
                                                         top10_overlap_with_wfst=round(overlap / max(1, n_over), 3),
                                                         ms_per_utterance_incl_host=round(dt / U * 1e3, 2))
        rows.append(row)
    return rows


def run():
    lib = N.load(); dev = torch.device("cuda:0"); _p = ops._p
    prons, words, arpa, g, seqs, logits, lens, build_s = make()
    U, T, C = logits.shape
    lg, pri, lp = _logp(logits, dev, lib)
    sys.stderr.write(f"TLG: {g.n_states} states, {g.n_arcs} arcs, built in {build_s:.1f} s; {U} x {T} frames\n")
    hs = int(os.environ.get("B2T_WFST_HASH", "0"))

    def timed(S, reps=3):
        ts, mem, fin = [], None, None
        for rep in range(reps):
            S.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
            N.check(lib.b2t_lm_prologue_f32(_p(lg), _p(pri), float(math.log(90.0)), _p(lp), U * T, C, ops._stream()), "prologue")
            S.search(lp, lens)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            mem = S.memory_stats()
            torch.cuda.synchronize(); t1b = time.perf_counter()
            fut = S.finalize_async()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            fin = fut.result(); t3 = time.perf_counter()
            ts.append((t1 - t0, t2 - t1b, t3 - t2))
        return [min(t[i] for t in ts) * 1e3 for i in range(3)] + [mem, fin]

    G_auto = lib.b2t_wfst_cluster_size(U)
    big = dict(max_frames=T + 8, max_tokens=1 << 22, max_links=1 << 24, hash_size=hs)
    # (a) round 2's form: one workgroup per utterance (32 of the 256 CUs), everything pruned once, at the end
    lib.b2t_wfst_set_cluster(1)
    S1 = WfstSearch(g, Opt, U=U, prune_interval=0, **big)
    search1_ms, _, _, _, _ = timed(S1, 2)
    del S1
    lib.b2t_wfst_set_cluster(0)
    # (b) clusters; PruneActiveTokens every 25 frames like the reference (lattice-faster-decoder.cc:592-630)
    Sr = WfstSearch(g, Opt, U=U, prune_interval=25, prune_min_fill=0.0, **big)
    search_ref_ms, fin_gpu_ref_ms, nbest_ref_ms, mem_ref, _ = timed(Sr, 2)
    del Sr
    # (c) clusters; the passes only when an utterance's arrays are half full (they only bound memory): the default here
    S = WfstSearch(g, Opt, U=U, prune_interval=25, prune_min_fill=0.5, **big)
    search_ms, fin_gpu_ms, nbest_ms, mem, fin = timed(S, 3)
    hdr = S._header()
    created_tok = float(sum(m["created_tokens"] for m in mem)); created_link = float(sum(m["created_links"] for m in mem))
    tok_per_frame = created_tok / float(hdr[:, 0].sum())
    # algorithmic bytes of the search: 16 B per expanded arc + 20 B per token + 21 B per forward link
    alg_bytes = 16.0 * sum(S.arcs_expanded()) + 20.0 * created_tok + 21.0 * created_link
    # (d) steady state over 6 batches with the host n-best of batch b running under the search of batch b + 1
    S2 = WfstSearch(g, Opt, U=U, prune_interval=25, prune_min_fill=0.5, **big)
    # a decode server freezes the interpreter's long-lived objects once it is set up: every n-best list is 10^4 new Python
    # objects, and the full garbage collection they trigger now and then walks every tracked object of the process (torch,
    # numpy, the graph: ~200 k) under the interpreter lock -- measured as one 60-90 ms finalize in ten without the freeze
    import gc
    gc.collect(); gc.freeze()
    # Two searchers on two streams.  Batch b + 1 is reset and searched (on the other stream, behind batch b's FinalizeDecoding
    # kernel: two cluster searches are never in flight together) BEFORE the host waits for batch b's lattices, so the lattice
    # extraction, the copy out and the host's share of batch b run under the next search instead of leaving the GPU idle
    # (22.9 ms per batch with one stream, where the GPU waited ~4 ms per batch for the host).
    st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
    st_a.wait_stream(torch.cuda.current_stream()); st_b.wait_stream(torch.cuda.current_stream())
    pend, t0, stamps = None, None, []
    NB = 10
    with torch.cuda.stream(st_a):
        S.reset(); S.search(lp, lens)
    for b in range(NB):
        if b == 2:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        cur, cur_st = (S, st_a) if b % 2 == 0 else (S2, st_b)
        nxt, nxt_st = (S2, st_b) if b % 2 == 0 else (S, st_a)
        ta = time.perf_counter()
        with torch.cuda.stream(cur_st):
            cur.finalize_begin()
        if b + 1 < NB:
            with torch.cuda.stream(nxt_st):
                nxt_st.wait_event(cur.finalize_event)
                nxt.reset(); nxt.search(lp, lens)
        tb = time.perf_counter()
        with torch.cuda.stream(cur_st):
            f = cur.finalize_collect()
        tc = time.perf_counter()
        if pend is not None:
            pend.result()
        pend = f
        stamps.append((ta, tb, tc, time.perf_counter()))
    fin_pipe = pend.result()
    torch.cuda.synchronize()
    pipelined_ms = (time.perf_counter() - t0) * 1e3 / (NB - 2)
    pipe_same = [[(e[2], round(e[3] + e[4], 3)) for e in u_] for u_ in fin_pipe] == [[(e[2], round(e[3] + e[4], 3)) for e in u_] for u_ in fin]
    # per batch: enqueue of FinalizeDecoding + the NEXT batch's reset and search, finalize_collect (waits for this batch's
    # lattices, copies them out), wait for the PREVIOUS batch's host n-best
    pipe_stages = [[round((y - x) * 1e3, 2) for x, y in zip(st[:-1], st[1:])] for st in stamps[2:]]
    del S2
    # the same search with one workgroup per utterance and every CU busy: 256 utterances in one call
    wide = None
    UW = int(os.environ.get("B2T_WFST_WIDE_U", "256"))
    if UW > U:
        try:
            rep = (UW + U - 1) // U
            lpw = lp.repeat(rep, 1, 1)[:UW].contiguous(); lensw = np.tile(lens, rep)[:UW]
            SW = WfstSearch(g, Opt, U=UW, max_frames=T + 8, max_tokens=1 << 21, max_links=1 << 23, hash_size=hs, prune_interval=0)
            tw = []
            for rep_ in range(2):
                SW.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
                SW.search(lpw, lensw); torch.cuda.synchronize(); tw.append(time.perf_counter() - t0)
            wide = dict(utterances=UW, workgroups_per_utterance=int(lib.b2t_wfst_cluster_size(UW)), search_ms=round(min(tw) * 1e3, 2),
                        search_ms_per_utterance=round(min(tw) * 1e3 / UW, 3), achieved_gb_s=round(alg_bytes * (UW / U) / min(tw) / 1e9, 1))
            del SW, lpw
            torch.cuda.empty_cache()
        except Exception as e:     # capacity of the box
            wide = dict(error=str(e)[:200])
    # Rescore() by lattice composition (brain_speech_decoder.cc:47-101) on the 32 lattices the search just left: the graph's 3-gram
    # out, an unpruned 4-gram over the same words in; 100-best by the new costs
    rescore = None
    try:
        word_id = {w: i for i, w in enumerate(g.words) if 0 < i <= len(words)}
        wd0 = g.words.index("#0")
        G_old = wfst.HostFst.from_fst(wfst.grammar_fst(arpa, word_id, wd0)).arcsort()
        G_new = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(words, 4, 6000, seed=77), word_id, wd0)).arcsort()
        t0 = time.perf_counter(); first = S._nbest_all(100); t1 = time.perf_counter()
        resc = S._nbest_all(100, rescore=(G_old, G_new, wd0)); t2 = time.perf_counter()
        again = []                                 # the same call twice more: the pool's threads keep their work arrays (graphc.cpp, DetRescore)
        for _ in range(2):
            ta = time.perf_counter(); S._nbest_all(100, rescore=(G_old, G_new, wd0)); again.append(time.perf_counter() - ta)
        changed = sum(1 for a, b in zip(first, resc) if a and b and a[0][2] != b[0][2])
        promoted = 0
        for a, b in zip(first, resc):
            seen = set(tuple(e[2]) for e in a)
            promoted += sum(1 for e in b[:10] if tuple(e[2]) not in seen)
        rescore = dict(nbest100_ms_32_utterances=round((t1 - t0) * 1e3, 2), rescore_nbest100_ms_32_utterances=round((t2 - t1) * 1e3, 2),
                       rescore_nbest100_ms_32_utterances_repeated=[round(x * 1e3, 2) for x in again], host_pool_threads=_pool_threads(), threads_per_lattice_of_60k_arcs_or_more=int(os.environ.get("B2T_RESCORE_BIG_THREADS", "4")),
                       utterances_whose_1best_changed=changed, top10_entries_from_below_the_first_100=promoted,
                       grammars="word 3-gram (in the graph) -> word 4-gram, 6000 n-grams per order")
    except Exception as e:     # noqa: BLE001
        rescore = dict(error=str(e)[:200])
    # streaming: one frame per call for all U utterances, partial best path read back; PruneActiveTokens every 25 frames
    # (prune_after_read: a pass that falls due is enqueued BEHIND the frame's partial result and runs while the host waits for the
    #  next frame -- 80 ms in real time; here the synchronize() in front of the timer stands for the idle GPU a frame arrives on, and
    #  the time it takes is reported as the pass's own: `prune_pass_ms_between_frames`.  `max_ms_per_frame_prune_inside` is the
    #  round-3 number: the pass inside the frame that made it due.)
    def stream_loop(S_, lp_, lens_, T_):
        lat_, between = [], []
        for t in range(T_):
            fr = lp_[:, t:t + 1].contiguous()
            tb = time.perf_counter(); torch.cuda.synchronize(); between.append(time.perf_counter() - tb)
            t0 = time.perf_counter()
            S_.search(fr, np.minimum(1, np.maximum(0, lens_ - t)).astype(np.int32))
            S_.best_path(False, max_len=2 * T_ + 8)
            lat_.append(time.perf_counter() - t0)
        return np.array(lat_[5:]) * 1e3, np.array(between[5:]) * 1e3
    Ss = WfstSearch(g, Opt, U=U, prune_interval=25, prune_min_fill=0.0, max_frames=T + 8, max_tokens=1 << 20, max_links=1 << 22, hash_size=hs)
    lat_inside, _ = stream_loop(Ss, lp, lens, T)
    del Ss
    Ss = WfstSearch(g, Opt, U=U, prune_interval=25, prune_min_fill=0.0, max_frames=T + 8, max_tokens=1 << 20, max_links=1 << 22, hash_size=hs,
                    prune_after_read=True)
    lat, lat_between = stream_loop(Ss, lp, lens, T)
    smem = Ss.memory_stats()
    Ss.finalize()
    del Ss
    # BASELINE configs[4] names a 5-gram LM: the same streaming loop over a WORD-LEVEL 5-gram graph (synthetic ARPA of the same
    # vocabulary, `n_per_order` n-grams per order as the recipe's pruned LM would keep; T o L o G by the same builder)
    stream5 = None
    try:
        t0 = time.time()
        arpa5 = ngram_lm.synthetic_word_arpa(words, 5, int(os.environ.get("B2T_WFST_5GRAM_PER_ORDER", "3000")), seed=12)
        g5 = wfst.build_tlg(prons, arpa5, sil_prob=0.5)
        b5 = time.time() - t0
        _, _, _, _, seqs5, logits5, lens5, _ = make(U=U, seed=5, noise=0.9, graph=(prons, words, arpa5, g5), truth="lm", blank_boost=math.log(90.0))
        _, _, lp5 = _logp(logits5, dev, lib)
        T5 = logits5.shape[1]
        S5 = WfstSearch(g5, Opt, U=U, prune_interval=25, prune_min_fill=0.0, max_frames=T5 + 8, max_tokens=1 << 20, max_links=1 << 22, hash_size=hs,
                        prune_after_read=True)
        lat5, lat5_between = stream_loop(S5, lp5, lens5, T5)
        fin5 = S5.finalize()
        h5 = [[g5.words[w] for w in f[0][2]] if f else [] for f in fin5]
        stream5 = dict(graph=dict(words=len(words), order=5, tlg_states=int(g5.n_states), tlg_arcs=int(g5.n_arcs), mb=round(g5.nbytes() / 1e6, 1), host_build_s=round(b5, 1)),
                       p50_ms_per_frame=round(float(np.percentile(lat5, 50)), 3), p95_ms_per_frame=round(float(np.percentile(lat5, 95)), 3),
                       max_ms_per_frame=round(float(lat5.max()), 3), prune_pass_ms_between_frames=round(float(lat5_between.max()), 3),
                       wer_vs_truth=round(sum(edit(h, r) for h, r in zip(h5, seqs5)) / max(1, sum(len(r) for r in seqs5)), 4))
        del S5
    except Exception as e:     # noqa: BLE001
        stream5 = dict(error=f"{type(e).__name__}: {str(e)[:200]}")
    wfst_1 = [[g.words[w] for w in f[0][2]] if f else [] for f in fin]
    err_truth = sum(edit(h, r) for h, r in zip(wfst_1, seqs)); nref = sum(len(r) for r in seqs)
    gbs = lambda ms: round(alg_bytes / (ms * 1e-3) / 1e9, 2)
    out = dict(graph=dict(words=len(words), tlg_states=int(g.n_states), tlg_arcs=int(g.n_arcs), mb=round(g.nbytes() / 1e6, 1),
                          host_build_s=round(build_s, 1)),
               offline=dict(utterances=U, frames=int(T), workgroups_per_utterance=int(G_auto), search_ms=round(search_ms, 2),
                            search_ms_prune_every_25_frames=round(search_ref_ms, 2),
                            search_ms_one_workgroup_per_utterance=round(search1_ms, 2),
                            finalize_gpu_ms=round(fin_gpu_ms, 2), nbest100_host_ms=round(nbest_ms, 2),
                            ms_per_utterance=round((search_ms + fin_gpu_ms + nbest_ms) / U, 3),
                            pipelined_ms_per_batch=round(pipelined_ms, 2), pipelined_ms_per_utterance=round(pipelined_ms / U, 3),
                            pipelined_stages_ms_enqueue_finalize_wait=pipe_stages, pipelined_lists_equal_one_shot=bool(pipe_same),
                            held_vs_created_tokens_with_pruning=round(sum(m["tokens"] for m in mem_ref) / max(1.0, created_tok), 3),
                            tokens_per_frame=round(tok_per_frame, 1), algorithmic_mb=round(alg_bytes / 1e6, 1),
                            achieved_gb_s=gbs(search_ms), achieved_gb_s_one_workgroup_per_utterance=gbs(search1_ms),
                            hbm_roofline_frac=round(alg_bytes / (search_ms * 1e-3) / 8.0e12, 5)),
               offline_all_cus=wide,
               streaming=dict(p50_ms_per_frame=round(float(np.percentile(lat, 50)), 3), p95_ms_per_frame=round(float(np.percentile(lat, 95)), 3),
                              max_ms_per_frame=round(float(lat.max()), 3), prune_pass_ms_between_frames=round(float(lat_between.max()), 3),
                              max_ms_per_frame_prune_inside=round(float(lat_inside.max()), 3),
                              held_tokens_at_end=int(max(m["tokens"] for m in smem)), created_tokens=int(max(m["created_tokens"] for m in smem)),
                              prune_passes=int(smem[0]["prunes"])),
               streaming_word_5gram=stream5,
               rescore=rescore,
               wfst_wer_vs_truth=round(err_truth / nref, 4))
    out["accuracy_by_noise"] = accuracy(prons, words, arpa, g)
    return out


if __name__ == "__main__":
    print(json.dumps(run(), indent=1))
