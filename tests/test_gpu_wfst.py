"""GPU: the batched WFST token-passing search (csrc/wfst.hip + wfst_decoder.py, through the C ABI) against the oracle's
restatement of the reference decoder (oracle/wfst_oracle.py) on the same graph and log-probabilities, with the production
options (language-model-standalone.py:486-496: beam 17, max_active 7000, min_active 200, lattice_beam 8, acoustic_scale
0.325, no blank skipping) and with blank skipping / tight pruning.  Costs: abs 2e-3 (fp32 sums in a different order);
word sequences and phoneme alignments identical wherever the oracle's cost gap to the next entry exceeds that."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import ngram_lm
import wfst
from oracle import wfst_oracle as W

# Round 6 (verdict item 6): the oracle is sequential Python and was 55 % of the GPU suite's wall time.  By default the oracle-bound
# cases run at reduced size (fewer utterances through the ORACLE; the HIP side keeps its full batch); B2T_TEST_FULL=1 restores
# every comparison of rounds 3-5.
FULL = os.environ.get("B2T_TEST_FULL", "0") == "1"

TOL = 2e-3


class Opt:
    def __init__(self, **kw):
        d = dict(max_active=7000, min_active=200, beam=17.0, lattice_beam=8.0, acoustic_scale=0.325,
                 ctc_blank_skip_threshold=1.0, length_penalty=0.0, nbest=20)
        d.update(kw)
        self.__dict__.update(d)


def cfg_of(o, cutoff_rule="sequential"):
    return W.Config(beam=o.beam, max_active=o.max_active, min_active=o.min_active, lattice_beam=o.lattice_beam,
                    acoustic_scale=o.acoustic_scale, nbest=o.nbest, blank_skip_thresh=o.ctc_blank_skip_threshold,
                    length_penalty=o.length_penalty, cutoff_rule=cutoff_rule)


@pytest.fixture(scope="module")
def toy():
    prons = ngram_lm.synthetic_lexicon(60, 41, seed=11)
    words = sorted(prons)
    arpa = ngram_lm.synthetic_word_arpa(words, 3, 400, seed=12)
    return prons, words, wfst.build_tlg(prons, arpa, sil_prob=0.5), arpa


def utterances(prons, words, U, rs, noise=1.2, blank_bias=math.log(90.0), n_words=(2, 5)):
    """Noisy log-probabilities spelling random word sequences, after the DecodeNumpy prologue (blank penalty)."""
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
    T = max(l.shape[0] for l in lps)
    batch = np.zeros((U, T, 41), np.float32)
    lens = np.array([l.shape[0] for l in lps], np.int32)
    for u, l in enumerate(lps):
        batch[u, :l.shape[0]] = l
    return seqs, lps, batch, lens


def compare_lists(got, ref_search, tag):
    """got: [(inputs, times, words, lm, ac)] from the HIP search; ref_search: the oracle's CtcWfstBeamSearch after finalize."""
    ref = list(zip(ref_search.inputs, ref_search.times, ref_search.outputs, ref_search.likelihood))
    assert len(got) == len(ref), (tag, len(got), len(ref))
    tot_ref = [-(l[0] + l[1]) for _, _, _, l in ref]
    for k, ((gi, gt, gw, glm, gac), (ri, rt, rw, (rlm, rac))) in enumerate(zip(got, ref)):
        assert abs((glm + gac) - (rlm + rac)) < TOL, (tag, k)
        gap_prev = tot_ref[k] - tot_ref[k - 1] if k > 0 else 1.0
        gap_next = tot_ref[k + 1] - tot_ref[k] if k + 1 < len(ref) else 1.0
        if min(gap_prev, gap_next) > 2 * TOL:          # unambiguous rank: same words, same graph / acoustic split, same alignment
            assert gw == list(rw), (tag, k)
            assert abs(glm - rlm) < TOL and abs(gac - rac) < TOL, (tag, k)
            assert gi == list(ri) and gt == list(rt), (tag, k)


def test_wfst_search_matches_oracle_production_options(toy):
    """8 utterances in one call, production options: partial best path after Search, n-best (20) after FinalizeSearch."""
    from wfst_decoder import WfstSearch
    prons, words, g, _ = toy
    rs = np.random.RandomState(3)
    seqs, lps, batch, lens = utterances(prons, words, 8, rs, noise=0.5, blank_bias=0.0)
    o = Opt()
    S = WfstSearch(g, o, U=8, max_frames=batch.shape[1] + 8)
    S.search(torch.from_numpy(batch).cuda(), lens)
    part = S.best_path(False)
    fin = S.finalize()
    hits = 0
    checked = range(8) if FULL else (0, 2, 3, 5, 7)
    for u in checked:
        R = W.CtcWfstBeamSearch(g, cfg_of(o))
        R.search(lps[u])
        assert S.frames_decoded()[u] == len(R.mapping)
        pi, pt, pw, plm, pac = part[u]
        assert pw == R.outputs[0] and pi == R.inputs[0]
        assert abs(plm - R.likelihood[0][0]) < TOL and abs(pac - R.likelihood[0][1]) < TOL
        R.finalize_search()
        compare_lists(fin[u], R, f"utt{u}")
        hits += [g.words[w] for w in fin[u][0][2]] == seqs[u]
    print(f"spelled sentences recovered exactly: {hits}/{len(checked)} (the LM weight and merged repeats change the others; the oracle agrees on all)")


def test_compact_arcs_equal_the_half_rounded_graph_bit_for_bit(toy):
    """Round 4: 10-byte arcs (labels = ilabel | olabel << 7 in one word, the weight as IEEE half; b2t_wfst_graph_t.compact).
    The search, the partial best path, FinalizeDecoding, the lattice and the n-best on the compact graph are those of the
    full-width graph whose weights were rounded to half beforehand -- bit for bit, in both search kernels, with pruning passes
    in between -- and the stated error bound of the rounding holds (the oracle legs above keep running on full-width arcs)."""
    import b2t_native as N
    from wfst_decoder import WfstSearch
    prons, words, g, _ = toy
    rs = np.random.RandomState(11)
    seqs, lps, batch, lens = utterances(prons, words, 6, rs, noise=0.7, blank_bias=0.0)
    o = Opt(nbest=30)
    import copy
    gc = copy.copy(g); gc._dev = None; gc.set_compact(True)
    gh = g.half_rounded()
    w = g.weight[np.isfinite(g.weight)]
    assert np.all(np.abs(gh.weight[np.isfinite(g.weight)] - w) <= np.abs(w) * 2.0 ** -11 + 1e-7)
    assert gc.nbytes() < g.nbytes() and (g.nbytes() - gc.nbytes()) == 6 * g.n_arcs
    dev_batch = torch.from_numpy(batch).cuda()
    lib = N.load()
    for cluster in (0, 1):
        lib.b2t_wfst_set_cluster(cluster)
        try:
            outs = []
            for graph in (gc, gh):
                S = WfstSearch(graph, o, U=6, max_frames=batch.shape[1] + 8, prune_interval=10, prune_min_fill=0.0)
                for t0 in range(0, batch.shape[1], 13):
                    chunk = dev_batch[:, t0:t0 + 13].contiguous()
                    S.search(chunk, np.clip(lens - t0, 0, chunk.shape[1]).astype(np.int32))
                part = S.best_path(False)
                fin = S.finalize()
                outs.append((S.frames_decoded(), part, fin))
            assert outs[0] == outs[1], f"compact arcs differ from the half-rounded graph (cluster setting {cluster})"
            assert any(len(f) > 3 for f in outs[0][2])
        finally:
            lib.b2t_wfst_set_cluster(0)


def test_wfst_streaming_equals_one_shot_and_blank_skipping(toy):
    """Chunk-by-chunk Search (state persists in HBM) == one call, bit for bit; blank-frame skipping with the re-insertion
    rule (ctc_wfst_beam_search.cc:79-94) reproduces the oracle's frame mapping, times and results."""
    from wfst_decoder import WfstSearch
    prons, words, g, _ = toy
    rs = np.random.RandomState(5)
    seqs, lps, batch, lens = utterances(prons, words, 4, rs, noise=0.6, blank_bias=0.0)
    for u in range(4):                         # make many frames near-certain blanks so that skipping happens
        blanks = np.argmax(lps[u], -1) == 0
        lps[u][blanks, 0] = math.log(0.995); batch[u, :lens[u]] = lps[u]
    o = Opt(ctc_blank_skip_threshold=0.98, nbest=10, acoustic_scale=0.5)
    dev_batch = torch.from_numpy(batch).cuda()
    A = WfstSearch(g, o, U=4, max_frames=batch.shape[1] + 8)
    A.search(dev_batch, lens)
    B = WfstSearch(g, o, U=4, max_frames=batch.shape[1] + 8)
    for t0 in range(0, batch.shape[1], 7):
        chunk = dev_batch[:, t0:t0 + 7].contiguous()
        B.search(chunk, np.clip(lens - t0, 0, chunk.shape[1]).astype(np.int32))
    assert A.frames_decoded() == B.frames_decoded()
    fa, fb = A.finalize(), B.finalize()
    assert fa == fb
    for u in range(4):
        R = W.CtcWfstBeamSearch(g, cfg_of(o))
        R.search(lps[u])
        assert len(R.mapping) < lens[u] and A.frames_decoded()[u] == len(R.mapping)
        R.finalize_search()
        compare_lists(fa[u], R, f"skip{u}")


def test_wfst_tight_pruning_and_overflow(toy):
    """max_active / min_active that bind (GetCutoff's nth_element branches): the best path still equals the oracle's;
    exhausted capacities are reported, never silently truncated."""
    from wfst_decoder import WfstSearch
    prons, words, g, _ = toy
    rs = np.random.RandomState(9)
    seqs, lps, batch, lens = utterances(prons, words, 3, rs, noise=1.0)
    o = Opt(max_active=300, min_active=50, beam=9.0, nbest=1, lattice_beam=4.0)
    S = WfstSearch(g, o, U=3, max_frames=batch.shape[1] + 8)
    S.search(torch.from_numpy(batch).cuda(), lens)
    fin = S.finalize()
    for u in range(3):
        R = W.CtcWfstBeamSearch(g, cfg_of(o))
        R.search(lps[u]); R.finalize_search()
        assert fin[u][0][2] == R.outputs[0]
        assert abs((fin[u][0][3] + fin[u][0][4]) - (R.likelihood[0][0] + R.likelihood[0][1])) < TOL
    small = WfstSearch(g, Opt(), U=1, max_frames=batch.shape[1] + 8, max_tokens=2000, max_links=4000)
    small.search(torch.from_numpy(batch[:1]).cuda(), lens[:1])
    with pytest.raises(RuntimeError, match="capacity"):
        small.finalize()


def test_search_on_the_determinised_minimised_graph(toy):
    """The graph as the reference's recipe leaves it (make_tlg.sh:43-46: L with lexicon disambiguation symbols o G,
    fstdeterminizestar --use-log, fstminimizeencoded, then T o LG; built by the native compiler, wfst.build_tlg_native): the
    disambiguation symbols become input-epsilon arcs, so the epsilon closure works harder than on the plain graph.  The HIP
    search on it equals the oracle on it (20-best lists), and its 1-best equals the plain graph's (same words; costs within
    the 1/1024-per-arc quantisation of minimize-encoded)."""
    from wfst_decoder import WfstSearch
    prons, words, g_plain, arpa = toy
    st = {}
    g_opt = wfst.build_tlg_native(prons, arpa, sil_prob=0.5, optimize=True, stats=st)
    assert g_opt.n_arcs < 0.8 * st["TLG"]["n_arcs"] or st["LG"]["n_states"] < st["LG_raw"]["n_states"]
    assert int((g_opt.n_eps > 0).sum()) > int((g_plain.n_eps > 0).sum())
    rs = np.random.RandomState(51)
    seqs, lps, batch, lens = utterances(prons, words, 4, rs, noise=1.0)
    o = Opt()
    out = {}
    for tag, g in (("opt", g_opt), ("plain", g_plain)):
        S = WfstSearch(g, o, U=4, max_frames=batch.shape[1] + 8)
        S.search(torch.from_numpy(batch).cuda(), lens)
        out[tag] = S.finalize()
    for u in range(4):
        if FULL or u != 2:
            R = W.CtcWfstBeamSearch(g_opt, cfg_of(o))
            R.search(lps[u]); R.finalize_search()
            compare_lists(out["opt"][u], R, f"optimised graph, utterance {u}")
        a, b = out["opt"][u][0], out["plain"][u][0]
        assert [g_opt.words[w] for w in a[2]] == [g_plain.words[w] for w in b[2]], u
        assert abs((a[3] + a[4]) - (b[3] + b[4])) < 0.05


def test_cluster_search_equals_single_workgroup(toy):
    """The search with 2 / 4 / 8 / 16 / 32 workgroups per utterance (clusters behind one XCD's L2: L2 atomics, sc1 loads, cluster
    barriers), and the same kernel with ONE member (what batches too wide for clusters run), returns what the single-workgroup
    kernel returns: frames, partial best path after every chunk, the 30-best
    list with both scores, alignment and times -- for the production options, with blank skipping and with binding
    max_active / min_active.  (Costs are compared to 1e-5: equal-cost ties aside, the arithmetic is the same.)"""
    import b2t_native as N
    from wfst_decoder import WfstSearch
    lib = N.load()
    assert [lib.b2t_wfst_cluster_size(u) for u in (1, 8, 32, 33, 64, 65, 128, 129, 256)] == [8, 8, 8, 4, 4, 2, 2, 1, 1]
    prons, words, g, _ = toy
    rs = np.random.RandomState(41)
    seqs, lps, batch, lens = utterances(prons, words, 6, rs, noise=1.0, n_words=(3, 7))
    dev_batch = torch.from_numpy(batch).cuda()
    cases = [Opt(nbest=30), Opt(nbest=30, ctc_blank_skip_threshold=0.9), Opt(nbest=30, max_active=250, min_active=60, beam=11.0)]
    try:
        for o in cases:
            out = {}
            for G in (1, -1, 2, 4, 8, 16, 32):     # 1: the single-workgroup kernel; -1: the cluster kernel with one member (wide batches)
                lib.b2t_wfst_set_cluster(G)
                assert lib.b2t_wfst_cluster_size(6) == abs(G)
                S = WfstSearch(g, o, U=6, max_frames=batch.shape[1] + 8, prune_interval=16)
                parts = []
                for t0 in range(0, batch.shape[1], 13):
                    S.search(dev_batch[:, t0:t0 + 13].contiguous(), np.clip(lens - t0, 0, 13))
                    parts.append([(p[2], round(p[3] + p[4], 3)) for p in S.best_path(False)])
                out[G] = (parts, S.frames_decoded(), S.finalize())
            for G in (-1, 2, 4, 8, 16, 32):
                assert out[G][1] == out[1][1] and out[G][0] == out[1][0], G
                for u in range(6):
                    a, b = out[G][2][u], out[1][2][u]
                    assert len(a) == len(b) > 0
                    for x, y in zip(a, b):
                        assert abs((x[3] + x[4]) - (y[3] + y[4])) < 1e-5, (G, u)
                        assert x[2] == y[2] and x[0] == y[0] and x[1] == y[1] and abs(x[3] - y[3]) < 1e-5, (G, u)
    finally:
        lib.b2t_wfst_set_cluster(0)


def test_deep_epsilon_fans_equal_the_oracle():
    """ProcessNonemitting (lattice-faster-decoder.cc:839-909) where the epsilon closure is deep and wide: a ternary tree of
    epsilon arcs, three levels deep, hung below the start state and below five mid-graph states, its 27 leaves re-entering the
    graph.  The cluster search closes a frame in ONE pass (whoever lowers a token relaxes its arcs, four pending tokens per
    thread at most): here a thread that pops a tree node pushes three children, so the stack overflows and the fallback rounds
    run as well.  Frames, partial path and n-best lists equal the oracle's for 8 / 1 cluster members and the single-workgroup
    kernel."""
    import b2t_native as N
    from wfst_decoder import WfstSearch
    lib = N.load()
    prons = ngram_lm.synthetic_lexicon(40, 41, seed=21)
    words = sorted(prons)
    arpa = ngram_lm.synthetic_word_arpa(words, 3, 300, seed=22)
    table = ["<eps>"] + words + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= len(words)}
    wd0, td0 = len(words) + 1, 42
    tok_prons = {w: [[int(c) + 1 for c in p] for p in ps] for w, ps in prons.items()}
    f = wfst.trim(wfst.compose(wfst.token_fst(40, [td0]), wfst.compose(wfst.lexicon_fst(tok_prons, word_id, 0.5, 2, td0, wd0),
                                                                     wfst.grammar_fst(arpa, word_id, wd0))))
    n_eps = np.zeros(f.n, np.int64)
    for a in f.arcs:
        n_eps[a[0]] += a[1] == 0
    rs = np.random.RandomState(3)
    no_eps = [s for s in range(f.n) if n_eps[s] == 0 and s != f.start]           # the leaves land here: no epsilon cycle arises
    with_eps = [s for s in range(f.n) if n_eps[s] > 0 and s != f.start]
    root = f.add_state()
    level = [root]
    for depth in range(3):
        nxt = []
        for i, s in enumerate(level):
            for k in range(3):
                d = f.add_state(); nxt.append(d)
                f.add_arc(s, 0, 0, 0.05 + 0.03 * ((7 * i + 3 * k + depth) % 5), d)
        level = nxt
    assert len(level) == 27
    for i, s in enumerate(level):
        for d in rs.choice(no_eps, 2, replace=False):
            f.add_arc(s, 0, 0, 0.1 + 0.02 * (i % 4), int(d))
    f.add_arc(f.start, 0, 0, 0.3, root)
    for s in rs.choice(with_eps, 5, replace=False):
        f.add_arc(int(s), 0, 0, 0.4, root)
    g = wfst.DecodeGraph(f, table)
    seqs, lps, batch, lens = utterances(prons, words, 3, np.random.RandomState(8), noise=1.0)
    dev_batch = torch.from_numpy(batch).cuda()
    o = Opt(nbest=15)
    refs, partial = [], []
    for u in range(3):
        R = W.CtcWfstBeamSearch(g, cfg_of(o)); R.search(lps[u])
        partial.append((len(R.mapping), list(R.outputs[0]), list(R.inputs[0])))
        R.finalize_search(); refs.append(R)
    try:
        for G in (0, -1, 1):
            lib.b2t_wfst_set_cluster(G)
            S = WfstSearch(g, o, U=3, max_frames=batch.shape[1] + 8)
            S.search(dev_batch, lens)
            part = S.best_path(False)
            for u in range(3):
                assert S.frames_decoded()[u] == partial[u][0]
                assert part[u][2] == partial[u][1] and part[u][0] == partial[u][2], (G, u)
            fin = S.finalize()
            for u in range(3):
                compare_lists(fin[u], refs[u], f"fan G={G} utt{u}")
    finally:
        lib.b2t_wfst_set_cluster(0)


def test_two_searchers_on_two_streams_pipeline(toy):
    """WfstPipeline (finalize_begin() / finalize_collect()): a second searcher on another stream starts the next batch behind
    the first one's FinalizeDecoding kernel (its `finalize_event`), so that lattice extraction, copy-out and host n-best of a
    batch run under the next batch's search.  Four batches alternating between two searchers return what finalize() returns
    for each."""
    from wfst_decoder import WfstSearch
    prons, words, g, _ = toy
    o = Opt(nbest=12)
    batches = []
    for seed in (51, 52, 53, 54):
        seqs, lps, batch, lens = utterances(prons, words, 5, np.random.RandomState(seed), noise=1.0)
        batches.append((torch.from_numpy(batch).cuda(), lens))
    T = max(b.shape[1] for b, _ in batches)
    ref = []
    R = WfstSearch(g, o, U=5, max_frames=T + 8)
    for b, lens in batches:
        R.reset(); R.search(b, lens); ref.append(R.finalize())
    from wfst_decoder import WfstPipeline
    pipe = WfstPipeline(g, o, U=5, max_frames=T + 8)
    assert list(pipe.decode(batches)) == ref
    assert list(pipe.decode(batches[:1])) == ref[:1] and list(pipe.decode([])) == []     # reusable; one batch; none


def test_prune_active_tokens_bounds_memory_and_keeps_the_lattice(toy):
    """PruneActiveTokens every prune_interval frames (lattice-faster-decoder.cc:516-545, :592-630) as b2t_wfst_prune between
    search calls: the n-best lists, the partial best paths and the decoded frames are those of a search that prunes only once,
    at the end (interval 0) -- for the reference's interval 25, for 7 (many passes, early stops, repeated compaction) and for a
    frame-by-frame stream -- while the tokens and links an utterance holds stay a fraction of what it created."""
    from wfst_decoder import WfstSearch
    prons, words, g, _ = toy
    rs = np.random.RandomState(31)
    seqs, lps, batch, lens = utterances(prons, words, 4, rs, noise=1.1, n_words=(9, 14))
    assert batch.shape[1] > 90
    dev_batch = torch.from_numpy(batch).cuda()
    o = Opt(nbest=30)
    runs = {}
    # ("stream25defer": prune_after_read -- the pass that falls due is enqueued behind the frame's partial result, or in front of the
    #  next search when nobody read one; the partial result is read every frame there, as a streaming consumer does)
    for tag, iv, stream in (("never", 0, False), ("ref25", 25, False), ("iv7", 7, False), ("stream25", 25, True), ("stream25defer", 25, True)):
        # (without the passes these four utterances overflow WfstSearch's default 512 k tokens: what the pruning is for)
        big = dict(max_tokens=1 << 21, max_links=1 << 23) if iv == 0 else {}
        S = WfstSearch(g, o, U=4, max_frames=batch.shape[1] + 8, prune_interval=iv, prune_after_read=tag.endswith("defer"), **big)
        parts = []
        if stream:
            for t in range(batch.shape[1]):
                S.search(dev_batch[:, t:t + 1].contiguous(), np.clip(lens - t, 0, 1))
                if tag.endswith("defer") and t % 3 != 2:
                    S.best_path(False)        # (two frames of three read their partial result: passes end up behind a read AND in front of a search)
                if t % 20 == 19 or t == batch.shape[1] - 1:
                    parts.append([p[2] for p in S.best_path(False)])
        else:
            for t0 in range(0, batch.shape[1], 20):
                S.search(dev_batch[:, t0:t0 + 20].contiguous(), np.clip(lens - t0, 0, 20))
                parts.append([p[2] for p in S.best_path(False)])
        mem = S.memory_stats()
        runs[tag] = (parts, S.frames_decoded(), mem, S.finalize())
    base = runs["never"]
    assert all(m["prunes"] == 0 for m in base[2])
    for tag in ("ref25", "iv7", "stream25", "stream25defer"):
        parts, frames, mem, fin = runs[tag]
        assert frames == base[1] and parts == base[0], tag
        for u in range(4):
            assert len(fin[u]) == len(base[3][u]) > 0, (tag, u)
            for a, b in zip(fin[u], base[3][u]):
                assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4], (tag, u)   # bit-identical
            assert mem[u]["prunes"] >= (3 if tag != "iv7" else 10)
            # held at the end (pruned lattice + the frames since the last pass) vs everything the search created
            assert mem[u]["tokens"] < 0.5 * base[2][u]["tokens"] and mem[u]["links"] < 0.5 * base[2][u]["links"], (tag, u, mem[u], base[2][u])
    # one utterance against the oracle too (which runs PruneActiveTokens every 25 frames itself)
    R = W.CtcWfstBeamSearch(g, cfg_of(o))
    R.search(lps[0]); R.finalize_search()
    compare_lists(runs["iv7"][3][0], R, "iv7 vs oracle")


def test_wfst_binding_regime_bench_graph():
    """The regime production runs in: the tools/bench_wfst.py graph (400 words x word 3-gram,
    120 k states / 419 k arcs), production options, 5 utterances with 8.3-8.6 k tokens per frame -- max_active = 7000 decides
    the cutoff in 335 of their 434 frames.  tests/golden/wfst_binding.npz holds what the sequential oracle produced (100-best
    lists) and, per frame, whether the over-the-cutoff tokens that only a sequential ProcessEmitting creates changed
    GetCutoff's result (21 of 434 frames: there the reference's own outcome depends on its hash-list traversal order).
    The HIP search must decode the same frames and return the same partial best path and the SAME 100-best list: order,
    words, graph / acoustic split, phoneme alignment and times -- also on the utterances with order-dependent frames
    (measured: the tokens in question lie far outside lattice_beam of anything that survives)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import wfst_binding_check as B
    Z, g, U = B.load()
    S, part, fin = B.run(Z, g, U)
    n_bound = n_dep = 0
    for u in range(U):
        st = Z[f"u{u}_stats"]
        assert st[:, 0].mean() > 7000 and st[:, 2].sum() > 0.5 * st.shape[0]      # the fixture IS the binding regime
        n_bound += int(st[:, 2].sum()); n_dep += int(st[:, 3].sum())
        assert S.frames_decoded()[u] == st.shape[0]
        assert list(part[u][2]) == Z[f"u{u}_partial_words"].tolist()
        ps = Z[f"u{u}_partial_scores"]
        assert abs((part[u][3] + part[u][4]) - (ps[0] + ps[1])) < TOL
        n = int(Z[f"u{u}_n"])
        assert len(fin[u]) == n == 100
        woff, aoff, sc = Z[f"u{u}_woff"], Z[f"u{u}_aoff"], Z[f"u{u}_scores"]
        tot_ref = -(sc[:, 0] + sc[:, 1])
        for k, (gi, gt, gw, glm, gac) in enumerate(fin[u]):
            assert abs(-(glm + gac) - tot_ref[k]) < 2e-4, (u, k)
            gap = min(tot_ref[k] - tot_ref[k - 1] if k else 1.0, tot_ref[k + 1] - tot_ref[k] if k + 1 < n else 1.0)
            if gap > 1e-3:                                   # rank unambiguous in fp32
                assert gw == Z[f"u{u}_words"][woff[k]:woff[k + 1]].tolist(), (u, k)
                assert abs(glm - sc[k, 0]) < 2e-4 and abs(gac - sc[k, 1]) < 2e-4, (u, k)
                assert gi == Z[f"u{u}_inputs"][aoff[k]:aoff[k + 1]].tolist() and gt == Z[f"u{u}_times"][aoff[k]:aoff[k + 1]].tolist(), (u, k)
    assert n_bound >= 300 and n_dep >= 10


def test_lm_decoder_surface_with_a_graph(toy, tmp_path):
    """The reference's call sequence (language-model-standalone.py:486-496,763-772; brain_speech_decoder.h:112-124):
    DecodeOptions(8 args) / DecodeResource(5 paths: TLG in the OpenFST vector container + words.txt) / BrainSpeechDecoder /
    DecodeNumpy per chunk / FinishDecoding / result()[i].sentence, .ac_score, .lm_score / Rescore / Reset."""
    import lm_decoder
    prons, words, g, arpa = toy
    f = wfst.Fst()
    f.n, f.start = g.n_states, g.start
    src = np.repeat(np.arange(g.n_states), np.diff(g.row))
    f.arcs = [(int(s), int(il), int(ol), float(w), int(d)) for s, il, ol, w, d in zip(src, g.ilabel, g.olabel, g.weight, g.next)]
    f.final = {int(s): float(c) for s, c in enumerate(g.final) if np.isfinite(c)}
    wfst.write_openfst_vector(f, str(tmp_path / "TLG.fst"))
    with open(str(tmp_path / "words.txt"), "w") as fh:
        for i, w in enumerate(g.words):
            fh.write(f"{w} {i}\n")
    opts = lm_decoder.DecodeOptions(7000, 200, 17.0, 8.0, 0.325, 1.0, 0.0, 10)
    res = lm_decoder.DecodeResource(str(tmp_path / "TLG.fst"), "", "", str(tmp_path / "words.txt"), "")
    dec = lm_decoder.BrainSpeechDecoder(res, opts)
    rs = np.random.RandomState(21)
    seqs, lps, batch, lens = utterances(prons, words, 1, rs, noise=0.8, blank_bias=0.0)
    logits = lps[0] + 1.7                                   # un-normalised, like the RNN's outputs
    for t0 in range(0, logits.shape[0], 10):                # streamed, one DecodeNumpy per chunk
        lm_decoder.DecodeNumpy(dec, logits[t0:t0 + 10], np.zeros_like(logits[t0:t0 + 10]), math.log(90.0))
        assert len(dec.result()) == 1
    assert dec.DecodedSomething()
    dec.FinishDecoding()
    out = dec.result()
    lp = logits - np.log(np.exp(logits).sum(-1, keepdims=True)); lp[:, 0] -= math.log(90.0)
    R = W.CtcWfstBeamSearch(g, W.Config(beam=17, max_active=7000, min_active=200, lattice_beam=8, acoustic_scale=0.325, nbest=10, blank_skip_thresh=1.0))
    R.search(lp.astype(np.float32)); R.finalize_search()
    want = W.decode_results(R, g.words)
    assert len(out) == len(want) and out[0].sentence == want[0][0]
    for r, (sent, ac, lm) in zip(out, want):
        assert abs(r.ac_score * 0.325 + r.lm_score - (ac * 0.325 + lm)) < TOL
    assert abs(out[0].ac_score - want[0][1]) < 1e-2 and abs(out[0].lm_score - want[0][2]) < TOL
    # Rescore with a second grammar: every sequence's graph score moves by G_old(W) - G_new(W); the list is re-ranked
    word_id = {w: i for i, w in enumerate(g.words) if 0 < i <= len(words)}
    wd0 = g.words.index("#0")
    G_old = wfst.grammar_fst(arpa, word_id, wd0)
    G_new = wfst.grammar_fst(ngram_lm.synthetic_word_arpa(words, 3, 400, seed=99), word_id, wd0)
    res.set_rescore_grammars(G_old, G_new, wd0)
    before = {r.sentence: (r.lm_score, r.ac_score) for r in out}
    dec.Rescore()
    after = dec.result()
    assert len(after) <= len(out) and len(after) >= 1
    tot = [r.lm_score + r.ac_score * 0.325 for r in after]
    assert all(a >= b - 1e-5 for a, b in zip(tot, tot[1:]))
    for r in after:
        if r.sentence in before:
            ids = r.word_ids
            delta = wfst.grammar_score(G_old, ids, wd0) - wfst.grammar_score(G_new, ids, wd0)
            assert abs(r.lm_score - (before[r.sentence][0] + delta)) < TOL and abs(r.ac_score - before[r.sentence][1]) < 1e-4
    # Rescore() composes the LATTICE with both grammars (b2t_lattice_rescore_nbest_host); the exchange on an exhaustive list of
    # the same lattice (the round-2 form, deep_list) must give the same ranking and scores
    after_scores = [(r.sentence, r.lm_score, r.ac_score) for r in after]
    dec.Rescore(deep_list=100000)
    listed = dec.result()
    assert len(listed) == len(after_scores)
    for (s1, lm1, ac1), r2 in zip(after_scores, listed):
        assert abs((lm1 + ac1 * 0.325) - (r2.lm_score + r2.ac_score * 0.325)) < TOL
    assert [s for s, _, _ in after_scores][0] == listed[0].sentence
    dec.Reset()
    assert dec.result() == [] and not dec.DecodedSomething()
    lm_decoder.DecodeNumpyLogProbs(dec, lp.astype(np.float32))
    dec.FinishDecoding()
    assert dec.result()[0].sentence == out[0].sentence
    # the reference's own way in (brain_speech_decoder.cc:61-101 via DecodeResource's five paths): both grammars as OpenFST
    # files, the back-off label taken from words.txt's "#0" -- must give what set_rescore_grammars gave
    wfst.write_openfst_vector(G_old, str(tmp_path / "G.fst"))
    wfst.write_openfst_vector(G_new, str(tmp_path / "G_no_prune.fst"))
    res2 = lm_decoder.DecodeResource(str(tmp_path / "TLG.fst"), str(tmp_path / "G.fst"), str(tmp_path / "G_no_prune.fst"),
                                     str(tmp_path / "words.txt"), "")
    # (the files carry #0:<eps> back-off arcs; ReadAndPrepareLmFst projects them on the output side: label 0 afterwards)
    assert res2.backoff_label == 0

    def rescored(res_x):
        dec2 = lm_decoder.BrainSpeechDecoder(res_x, opts)
        lm_decoder.DecodeNumpy(dec2, logits, np.zeros_like(logits), math.log(90.0))
        dec2.FinishDecoding()
        dec2.Rescore()
        return dec2.result()

    def same(got2):
        assert [r.sentence for r in got2] == [r.sentence for r in after]
        for r2, r1 in zip(got2, after):
            assert abs(r2.lm_score - r1.lm_score) < TOL and abs(r2.ac_score - r1.ac_score) < 1e-4

    same(rescored(res2))
    # grammars that ALREADY are acceptors with epsilon back-off arcs (projected beforehand, or compiled without eps2disambig),
    # next to a words.txt that lists "#0" as Kaldi's always do: the label comes from the arcs, not from the table (advisor,
    # round 3: with id("#0") no sequence that needs a back-off survived and Rescore() silently returned fewer hypotheses)
    def variant(G, mapper):
        H = wfst.Fst(); H.n, H.start, H.final = G.n, G.start, dict(G.final)
        H.arcs = [mapper(a) for a in G.arcs]
        return H
    proj = lambda a: (a[0], a[2], a[2], a[3], a[4])
    wfst.write_openfst_vector(variant(G_old, proj), str(tmp_path / "Ga.fst"))
    wfst.write_openfst_vector(variant(G_new, proj), str(tmp_path / "Gb.fst"))
    res3 = lm_decoder.DecodeResource(str(tmp_path / "TLG.fst"), str(tmp_path / "Ga.fst"), str(tmp_path / "Gb.fst"), str(tmp_path / "words.txt"), "")
    assert res3.backoff_label == 0
    same(rescored(res3))
    # #0 on BOTH sides of the back-off arcs (an acceptor no arc of which carries epsilon): followed through id("#0")
    both = lambda a: (a[0], a[1], a[1], a[3], a[4])
    wfst.write_openfst_vector(variant(G_old, both), str(tmp_path / "Gc.fst"))
    wfst.write_openfst_vector(variant(G_new, both), str(tmp_path / "Gd.fst"))
    res4 = lm_decoder.DecodeResource(str(tmp_path / "TLG.fst"), str(tmp_path / "Gc.fst"), str(tmp_path / "Gd.fst"), str(tmp_path / "words.txt"), "")
    assert res4.backoff_label == wd0
    same(rescored(res4))


def test_lattice_nbest_host_against_enumeration():
    """b2t_lattice_nbest_host (csrc/lattice.cpp) on a small random acyclic lattice: the cheapest path of every distinct word
    sequence, in order, with its graph / acoustic split and alignment -- against exhaustive enumeration."""
    import ctypes as C
    import b2t_native as N
    lib = N.load()
    rs = np.random.RandomState(4)
    n = 14
    src, dst, il, ol, gr, ac = [], [], [], [], [], []
    for s in range(n - 1):
        for _ in range(3):
            d = rs.randint(s + 1, min(n, s + 4))
            src.append(s); dst.append(d); il.append(rs.randint(0, 5)); ol.append(int(rs.choice([0, 0, 7, 8, 9])))
            gr.append(float(rs.rand())); ac.append(float(rs.rand() * 2))
    A = [np.array(x, dtype=t) for x, t in ((src, np.int32), (dst, np.int32), (il, np.int32), (ol, np.int32), (gr, np.float32), (ac, np.float32))]
    fs, fc = np.array([n - 1, n - 2], np.int32), np.array([0.25, 1.0], np.float32)
    best = {}
    stack = [(0, (), (), 0.0, 0.0)]
    while stack:
        s, w, a, g_, a_ = stack.pop()
        for k, st in enumerate(fs):
            if s == st:
                c = g_ + a_ + float(fc[k])
                if c < best.get(w, (math.inf,))[0]:
                    best[w] = (c, g_ + float(fc[k]), a_, a)
        for i in range(len(src)):
            if src[i] == s:
                stack.append((dst[i], w + ((ol[i],) if ol[i] else ()), a + ((il[i],) if il[i] else ()), g_ + gr[i], a_ + ac[i]))
    ranked = sorted(best.items(), key=lambda kv: kv[1][0])
    nb, beam = 6, 2.5
    ranked = [r for r in ranked if r[1][0] <= ranked[0][1][0] + beam][:nb]
    ow, oa = np.zeros(256, np.int32), np.zeros(256, np.int32)
    woff, aoff, costs = np.zeros(nb + 1, np.int32), np.zeros(nb + 1, np.int32), np.zeros(2 * nb, np.float32)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    k = lib.b2t_lattice_nbest_host(n, 0, len(src), P(A[0]), P(A[1]), P(A[2]), P(A[3]), P(A[4]), P(A[5]), 2, P(fs), P(fc), nb,
                                   C.c_float(beam), P(ow), P(woff), 256, P(oa), P(aoff), 256, P(costs))
    assert k == len(ranked) and k >= 2
    for j, (w, (c, g_, a_, a)) in enumerate(ranked):
        assert tuple(ow[woff[j]:woff[j + 1]]) == w
        assert abs(costs[2 * j] - g_) < 1e-5 and abs(costs[2 * j + 1] - a_) < 1e-5
        assert tuple(oa[aoff[j]:aoff[j + 1]]) == a


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_wfst_search_matches_oracle_on_random_graphs_and_options(seed):
    """Fuzz: a random lexicon (20-50 words), a random 2- or 3-gram, random noise and random options (beam, max_active small enough to
    bind or not, blank skipping, length penalty, prune interval, compact arcs, streamed frame by frame or in one call) -- n-best lists
    against the oracle's (oracle/wfst_oracle.py; unpinned itself, as everywhere in this file).  max_active stays >= 400 HERE; the
    regime in which the cut-off binds in every frame (max_active 60 / 150 on these graphs), where the reference's sequential
    next_cutoff tightening depends on the order of its hash list, is the next test's."""
    from wfst_decoder import WfstSearch
    rs = np.random.RandomState(seed)
    for case in range(3):
        n_words = int(rs.randint(20, 51)); order = int(rs.randint(2, 4))
        prons = ngram_lm.synthetic_lexicon(n_words, 41, seed=seed * 10 + case)
        words = sorted(prons)
        arpa = ngram_lm.synthetic_word_arpa(words, order, int(rs.randint(80, 300)), seed=seed * 10 + case + 1)
        g = wfst.build_tlg(prons, arpa, sil_prob=float(rs.choice([0.3, 0.5, 0.7])))
        g_dev = g
        if rs.rand() < 0.4:     # compact arcs: fp16 weights -- the oracle walks the same graph with its weights rounded to fp16
            import copy
            g = g.half_rounded()
            g_dev = copy.copy(g); g_dev._dev = None; g_dev.set_compact(True)
        U = int(rs.randint(1, 5))
        seqs, lps, batch, lens = utterances(prons, words, U, rs, noise=float(rs.choice([0.4, 0.9, 1.4])), blank_bias=float(rs.choice([0.0, math.log(90.0)])),
                                            n_words=(1, 4))
        o = Opt(beam=float(rs.choice([8.0, 12.0, 17.0])), max_active=int(rs.choice([400, 7000])), min_active=int(rs.choice([0, 20, 200])),
                lattice_beam=float(rs.choice([4.0, 8.0])), ctc_blank_skip_threshold=float(rs.choice([1.0, 0.98])),
                length_penalty=float(rs.choice([0.0, -0.3])), nbest=int(rs.choice([5, 20])))
        if o.min_active > o.max_active:
            o.min_active = 0
        iv = int(rs.choice([0, 7, 25]))
        S = WfstSearch(g_dev, o, U=U, max_frames=batch.shape[1] + 8, prune_interval=iv, prune_after_read=bool(rs.rand() < 0.5))
        dev_batch = torch.from_numpy(batch).cuda()
        if rs.rand() < 0.5:
            for t in range(batch.shape[1]):
                S.search(dev_batch[:, t:t + 1].contiguous(), np.clip(lens - t, 0, 1))
                if t % 5 == 4:
                    S.best_path(False)
        else:
            S.search(dev_batch, lens)
        fin = S.finalize()
        for u in range(U):
            # lists against the oracle's data-parallel cut-off rule (what csrc/wfst.hip computes: exact in every regime); the best
            # hypothesis also against the reference-order walk -- on these small graphs max_active = 400 can bind, and the two rules
            # then differ in the tail of a list (tools/r5_wfst_fuzz_more.py: 2 of 12 further seeds, ranks 14 and 18; NOTES.md R5.2)
            tag = f"seed {seed} case {case} utt {u}: words {n_words} order {order} opts {o.__dict__} interval {iv}"
            R = W.CtcWfstBeamSearch(g, cfg_of(o, "final"))
            R.search(lps[u]); R.finalize_search()
            compare_lists(fin[u], R, tag)
            Rs = W.CtcWfstBeamSearch(g, cfg_of(o, "sequential"))
            Rs.search(lps[u]); Rs.finalize_search()
            d = _first_diff(fin[u], Rs)
            assert d != 0, tag + ": best hypothesis differs from the reference-order oracle's"
            if d > 0:      # (not an error: where the data-parallel and the reference-order cut-off rules part in this list -- printed with -s)
                print(f"cut-off rules part at rank {d} of {len(fin[u])}: {tag}")



def _first_diff(got, ref_search):
    """first rank at which the HIP list and the oracle's differ (word sequence, or total cost by more than TOL); -1: none"""
    ref = list(zip(ref_search.outputs, ref_search.likelihood))
    for k, ((gi, gt, gw, glm, gac), (rw, (rlm, rac))) in enumerate(zip(got, ref)):
        if gw != list(rw) or abs((glm + gac) - (rlm + rac)) > TOL:
            return k
    return -1 if len(got) == len(ref) else min(len(got), len(ref))


@pytest.mark.parametrize("max_active", [60, 150])
def test_wfst_search_while_max_active_binds_in_every_frame(max_active):
    """Round-5 verdict 1d: the regime the fuzz above used to exclude.  The reference's ProcessEmitting tightens next_cutoff while it
    walks its hash list (lattice-faster-decoder.cc:786-810), so which tokens beyond the frame's FINAL cutoff get created depends on
    the list's order (kaldi/util/hash-list-inl.h), and the next frame's GetCutoff counts them (:650-720).  The oracle restates both
    readings: cutoff_rule="sequential" walks in HashList order (the reference), "final" compares every candidate with the frame's
    final cutoff (the data-parallel rule of csrc/wfst.hip).  Contract on the nine fuzz graphs with max_active 60 / 150:
      * HIP == oracle("final"): identical n-best lists (compare_lists) -- the kernel is pinned to a precise rule in this regime too;
      * HIP vs oracle("sequential"): the best hypothesis (words, cost) is identical in every utterance; lists may differ in their
        tail: measured on these 21 utterances (tools/r5_cutoff_order.py, CPU): max_active 60 -> 1 list differs, from rank 4 on
        (lengths 20 / 20); max_active 150 -> none.  The per-utterance table is printed."""
    from wfst_decoder import WfstSearch
    n_utt = n_tail = 0
    for seed in (101, 202, 303):
        rs = np.random.RandomState(seed)
        for case in range(3):
            n_words = int(rs.randint(20, 51)); order = int(rs.randint(2, 4))
            prons = ngram_lm.synthetic_lexicon(n_words, 41, seed=seed * 10 + case)
            words = sorted(prons)
            arpa = ngram_lm.synthetic_word_arpa(words, order, int(rs.randint(80, 300)), seed=seed * 10 + case + 1)
            g = wfst.build_tlg(prons, arpa, sil_prob=float(rs.choice([0.3, 0.5, 0.7])))
            U = int(rs.randint(1, 5))
            seqs, lps, batch, lens = utterances(prons, words, U, rs, noise=float(rs.choice([0.4, 0.9, 1.4])), blank_bias=float(rs.choice([0.0, math.log(90.0)])),
                                                n_words=(1, 4))
            o = Opt(beam=float(rs.choice([8.0, 12.0, 17.0])), max_active=max_active, min_active=int(rs.choice([0, 20, 200])),
                    lattice_beam=float(rs.choice([4.0, 8.0])), ctc_blank_skip_threshold=float(rs.choice([1.0, 0.98])),
                    length_penalty=float(rs.choice([0.0, -0.3])), nbest=int(rs.choice([5, 20])))
            if o.min_active > o.max_active:
                o.min_active = 0
            S = WfstSearch(g, o, U=U, max_frames=batch.shape[1] + 8, prune_interval=int(rs.choice([0, 7, 25])))
            S.search(torch.from_numpy(batch).cuda(), lens)
            fin = S.finalize()
            for u in range(U):
                tag = f"max_active {max_active} seed {seed} case {case} utt {u}"
                Rf = W.CtcWfstBeamSearch(g, cfg_of(o, "final"))
                Rf.search(lps[u]); Rf.finalize_search()
                compare_lists(fin[u], Rf, tag + " (final-cutoff rule)")
                Rs = W.CtcWfstBeamSearch(g, cfg_of(o, "sequential"))
                Rs.search(lps[u]); Rs.finalize_search()
                k = _first_diff(fin[u], Rs)
                print(f"{tag}: HIP list {len(fin[u])}, reference-order oracle {len(Rs.outputs)}, first differing rank {'none' if k < 0 else k}")
                assert k != 0, tag + ": the best hypothesis differs from the reference-order oracle's"
                n_utt += 1; n_tail += k > 0
    print(f"max_active {max_active}: {n_utt} utterances, {n_tail} whose list differs from the reference-order oracle's beyond rank 0")
    assert n_tail <= 2


def test_wfst_streamed_on_a_word_5gram_graph_32_utterances():
    """BASELINE configs[4] (language-model-standalone.py:486-496 with the 5-gram LM): 32 concurrent utterances streamed one frame per
    call over T o L o G of a WORD 5-gram (60 words, 300 n-grams per order: 24.6 k states / 81 k arcs), production options, PruneActiveTokens
    every 10 frames behind the frame's partial result, the running best path read every 4th frame -- partial results and final
    20-best lists against the oracle (reference order; max_active does not bind here), utterance by utterance."""
    from wfst_decoder import WfstSearch
    prons = ngram_lm.synthetic_lexicon(60, 41, seed=21)
    words = sorted(prons)
    arpa = ngram_lm.synthetic_word_arpa(words, 5, 300, seed=22)
    g = wfst.build_tlg(prons, arpa, sil_prob=0.5)
    U = 32
    rs = np.random.RandomState(3)
    seqs, lps, batch, lens = utterances(prons, words, U, rs, noise=0.9, n_words=(1, 3))
    o = Opt()
    S = WfstSearch(g, o, U=U, max_frames=batch.shape[1] + 8, prune_interval=10, prune_after_read=True)
    dev_batch = torch.from_numpy(batch).cuda()
    T = batch.shape[1]
    # (all 32 are streamed on the GPU; by default every third one also through the oracle -- B2T_TEST_FULL=1: all of them)
    chk = [u for u in range(U) if FULL or u % 3 == 0]
    R = {u: W.CtcWfstBeamSearch(g, cfg_of(o)) for u in chk}
    for t in range(T):
        S.search(dev_batch[:, t:t + 1].contiguous(), np.clip(lens - t, 0, 1))
        for u in chk:
            if t < lens[u]:
                R[u].search(lps[u][t:t + 1])
        if t % 4 == 3:
            part = S.best_path(False)
            for u in chk:
                pi, pt, pw, plm, pac = part[u]
                assert pw == R[u].outputs[0] and pi == R[u].inputs[0], (t, u)
                assert abs(plm - R[u].likelihood[0][0]) < TOL and abs(pac - R[u].likelihood[0][1]) < TOL, (t, u)
    fd = list(S.frames_decoded())
    assert [fd[u] for u in chk] == [len(R[u].mapping) for u in chk]
    fin = S.finalize()
    hits = 0
    for u in range(U):
        if u in R:
            R[u].finalize_search()
            compare_lists(fin[u], R[u], f"5-gram stream utt {u}")
        hits += [g.words[w] for w in fin[u][0][2]] == seqs[u]
    print(f"5-gram graph, 32 streamed utterances ({len(chk)} of them against the oracle): {hits}/32 spelled sentences recovered exactly")


def test_cluster_finalize_equals_single_workgroup_finalize(toy, monkeypatch):
    """Round 5 (verdict item 4): FinalizeDecoding by the utterance's cluster -- L2 atomics on the extra costs, one cluster barrier per
    frame + one per epsilon sweep, every link's alive byte written once -- against the one-workgroup kernel (`B2T_WFST_FIN_CLUSTER=0`)
    on the SAME search state (the state block is snapshotted behind the search and restored: two searches number their tokens in
    arrival order, i.e. differently): the pruned lattice (states, arcs, labels, both costs, final costs) is identical array by array,
    and so are the n-best lists -- production options, blank skipping, binding max_active, a streamed utterance with PruneActiveTokens
    passes in between, an utterance of a single frame."""
    from wfst_decoder import WfstSearch
    prons, words, g, _ = toy
    rs = np.random.RandomState(77)
    seqs, lps, batch, lens = utterances(prons, words, 8, rs, noise=1.0, n_words=(1, 6))
    lens = lens.copy(); lens[7] = 1                      # F = 1: the last frame is also (almost) the first
    dev_batch = torch.from_numpy(batch).cuda()
    for o, interval in ((Opt(nbest=30), 0), (Opt(nbest=30, ctc_blank_skip_threshold=0.9), 0), (Opt(nbest=30, max_active=250, min_active=60, beam=11.0), 0),
                        (Opt(nbest=30, lattice_beam=4.0), 9)):
        S = WfstSearch(g, o, U=8, max_frames=batch.shape[1] + 8, prune_interval=interval)
        if interval:
            for t in range(batch.shape[1]):
                S.search(dev_batch[:, t:t + 1].contiguous(), np.clip(lens - t, 0, 1))
        else:
            S.search(dev_batch, lens)
        torch.cuda.synchronize()
        snap = S.state.clone()
        res = {}
        for mode in ("0", "1", "1"):                     # (the cluster kernel twice: its barrier counters carry over in the block)
            monkeypatch.setenv("B2T_WFST_FIN_CLUSTER", mode)
            S.state.copy_(snap)
            fin = S.finalize()
            cn, (arrs, a_off, f_off) = S._lattices()
            got = (fin, np.array(cn).copy(), [np.array(a).copy() for a in arrs], np.array(a_off).copy(), np.array(f_off).copy())
            if mode in res:
                a, b = res[mode], got
            else:
                res[mode] = got
                if mode == "0":
                    continue
                a, b = res["0"], got
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]), o.__dict__
            assert int(a[1][:, 1].sum()) > 1000                    # (arcs survive: the comparison is not of empty lattices)
            for x, y, name in zip(a[2], b[2], ("src", "dst", "ilabel", "olabel", "graph", "acoustic", "final_state", "final_cost")):
                assert np.array_equal(x, y), (name, o.__dict__)
            assert a[0] == b[0], o.__dict__
    monkeypatch.delenv("B2T_WFST_FIN_CLUSTER", raising=False)


def test_cluster_prune_equals_single_workgroup_prune(toy, monkeypatch):
    """Round 5 (verdict item 4): PruneActiveTokens by the utterance's cluster -- the per-frame walk with L2 atomics, the stable
    compaction with one cluster barrier per 8 x 4096 elements -- against the one-workgroup pass (`B2T_WFST_PRUNE_CLUSTER=0`) on the SAME
    search state (snapshot / restore, as in the finalize test): after the pass the state block's token and link arrays, the frame
    offsets and the header counters are identical; decoding then continues from both and ends in identical lattices.  Several passes
    in a row (a pass over an already pruned lattice stops after a few frames), delta 0 and the production delta, an utterance that
    ended long ago next to running ones."""
    import b2t_native as N
    import ctypes as C
    from wfst_decoder import WfstSearch
    lib = N.load()
    prons, words, g, _ = toy
    rs = np.random.RandomState(91)
    seqs, lps, batch, lens = utterances(prons, words, 8, rs, noise=1.0, n_words=(1, 6))
    lens = lens.copy(); lens[6] = 3
    dev_batch = torch.from_numpy(batch).cuda()
    T = batch.shape[1]
    for o, delta in ((Opt(nbest=30), None), (Opt(nbest=30, lattice_beam=4.0), 0.0), (Opt(nbest=30, max_active=250, min_active=60, beam=11.0), None)):
        S = WfstSearch(g, o, U=8, max_frames=T + 8, prune_interval=0)
        # b2t_wfst_state_offsets: h, mapping, tok_off, link_off, cost_offset, tok_state, tok_cost, tok_extra, link_src, link_dst, link_arc, link_ac, link_graph, link_alive, tok_best, last_prob
        off = S.off
        arrays = dict(tok_off=(2, 4), link_off=(3, 4), tok_state=(5, 4), tok_cost=(6, 4), tok_extra=(7, 4), link_src=(8, 4), link_dst=(9, 4), link_arc=(10, 4),
                      link_ac=(11, 4), link_graph=(12, 4), link_alive=(13, 1), tok_best=(14, 8))
        d = float(o.lattice_beam) * 0.1 if delta is None else delta
        cuts = [0, 11, 30, 31, 55, T]
        for a, b in zip(cuts[:-1], cuts[1:]):
            S.search(dev_batch[:, a:b].contiguous(), np.clip(lens - a, 0, b - a))
            torch.cuda.synchronize()
            snap = S.state.clone()
            outs = []
            for mode in ("0", "1"):
                monkeypatch.setenv("B2T_WFST_PRUNE_CLUSTER", mode)
                S.state.copy_(snap)
                N.check(lib.b2t_wfst_prune(C.byref(S.cg), C.byref(S.co), ops_p(S.state), 8, C.c_float(d), C.c_float(0.0), S._s()), "b2t_wfst_prune")
                torch.cuda.synchronize()
                outs.append(S.state.clone())
            v0, v1 = (x.view(8, S.state_bytes) for x in outs)
            hdr0, hdr1 = (v[:, :80].contiguous().view(torch.int32).cpu().numpy() for v in (v0, v1))
            assert np.array_equal(hdr0, hdr1), (a, b, hdr0[:, :4], hdr1[:, :4])
            n_tok, n_link, nfr = hdr0[:, 1], hdr0[:, 2], hdr0[:, 0]
            assert int(hdr0[:, 17].sum()) > 0 or a == 0          # (removed_link: the passes do remove something)
            for u in range(8):
                for name, (k, wd) in arrays.items():
                    n = {"tok_off": nfr[u] + 2, "link_off": 2 * nfr[u] + 2}.get(name, n_tok[u] if name.startswith("tok_") else n_link[u])
                    x0 = v0[u, off[k]:off[k] + wd * int(n)].cpu().numpy(); x1 = v1[u, off[k]:off[k] + wd * int(n)].cpu().numpy()
                    assert np.array_equal(x0, x1), (name, u, a, b, o.__dict__)
            # (decoding goes on from the cluster pass's state)
        fin = S.finalize()
        # ... and ends where a searcher ends that took the same passes with the one-workgroup kernel (lists, not arrays: two searches
        # number their tokens in arrival order).  Against the ORACLE only the first case is compared: with delta 0 / a 4.0 lattice beam
        # a pass at an arbitrary frame can drop a path that sits on the beam's edge in fp32 -- either kernel, same lists.
        monkeypatch.setenv("B2T_WFST_PRUNE_CLUSTER", "0")
        S0 = WfstSearch(g, o, U=8, max_frames=T + 8, prune_interval=0)
        for a, b in zip(cuts[:-1], cuts[1:]):
            S0.search(dev_batch[:, a:b].contiguous(), np.clip(lens - a, 0, b - a))
            N.check(lib.b2t_wfst_prune(C.byref(S0.cg), C.byref(S0.co), ops_p(S0.state), 8, C.c_float(d), C.c_float(0.0), S0._s()), "b2t_wfst_prune")
        fin0 = S0.finalize()
        for u in range(8):
            assert len(fin[u]) == len(fin0[u]) > 0
            for x, y in zip(fin[u], fin0[u]):
                assert x[2] == y[2] and x[0] == y[0] and x[1] == y[1] and abs(x[3] - y[3]) < 1e-5 and abs(x[4] - y[4]) < 1e-5, (u, o.__dict__)
        if delta is None and o.max_active > 1000:
            for u in range(8):
                R = W.CtcWfstBeamSearch(g, cfg_of(o))
                R.search(lps[u][:lens[u]]); R.finalize_search()
                compare_lists(fin[u], R, f"after cluster prune passes, utt {u}")
    monkeypatch.delenv("B2T_WFST_PRUNE_CLUSTER", raising=False)


def ops_p(t):
    import b2t_ops as ops
    return ops._p(t)
