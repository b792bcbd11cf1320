"""Host n-best of the pruned token lattice (csrc/lattice.cpp: b2t_lattice_nbest_host -- FinalizeSearch's
DeterminizeLatticePruned + ShortestPath(nbest), ctc_wfst_beam_search.cc:123-160) against the oracle's restatement
(oracle/wfst_oracle.py: nbest_word_sequences) -- no GPU involved.  Random acyclic lattices with epsilon-output arcs, and a
real lattice the GPU search left behind for one utterance of the tools/bench_wfst.py workload (tests/golden/wfst_lattice_u2.npz,
written by attic/dump_lattice.py: 4868 states, 9329 arcs, 217 final states)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import b2t_native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import wfst_oracle as O      # noqa: E402


def host_nbest(n_states, start, src, dst, il, ol, gr, ac, fs, fc, nbest, beam, cap=1 << 16):
    lib = N.load()
    A = [np.ascontiguousarray(x, dtype=t) for x, t in ((src, np.int32), (dst, np.int32), (il, np.int32), (ol, np.int32), (gr, np.float32), (ac, np.float32))]
    fs, fc = np.ascontiguousarray(fs, np.int32), np.ascontiguousarray(fc, np.float32)
    ow, oa = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    woff, aoff, costs = np.zeros(nbest + 1, np.int32), np.zeros(nbest + 1, np.int32), np.zeros(2 * nbest, np.float32)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    k = lib.b2t_lattice_nbest_host(n_states, start, len(A[0]), P(A[0]), P(A[1]), P(A[2]), P(A[3]), P(A[4]), P(A[5]), len(fs), P(fs), P(fc), nbest,
                                   C.c_float(beam), P(ow), P(woff), cap, P(oa), P(aoff), cap, P(costs))
    assert k >= 0, N.last_error()
    return [(tuple(ow[woff[j]:woff[j + 1]].tolist()), float(costs[2 * j]), float(costs[2 * j + 1]), tuple(oa[aoff[j]:aoff[j + 1]].tolist())) for j in range(k)]


def oracle_nbest(n_states, start, src, dst, il, ol, gr, ac, fs, fc, nbest, beam):
    arcs = [[] for _ in range(n_states)]
    for i in range(len(src)):
        arcs[int(src[i])].append((int(il[i]), int(ol[i]), float(np.float32(gr[i])), float(np.float32(ac[i])), int(dst[i])))
    finals = {}
    for s, c in zip(fs, fc):
        finals[int(s)] = min(finals.get(int(s), float("inf")), float(np.float32(c)))
    return [(tuple(w), g, a, tuple(ali)) for (_tot, g, a, w, ali) in O.nbest_word_sequences(arcs, finals, start, nbest, beam)]


def compare(got, want):
    assert len(got) == len(want) and len(got) > 0
    # same costs in the same order; words and alignments equal wherever the neighbouring totals are not tied
    tot_g = np.array([g + a for _, g, a, _ in got]); tot_w = np.array([g + a for _, g, a, _ in want])
    np.testing.assert_allclose(tot_g, tot_w, rtol=0, atol=2e-4)
    for j, (gw, ww) in enumerate(zip(got, want)):
        tied = (j > 0 and abs(tot_w[j] - tot_w[j - 1]) < 1e-4) or (j + 1 < len(want) and abs(tot_w[j + 1] - tot_w[j]) < 1e-4)
        if tied:
            continue
        assert gw[0] == ww[0], f"entry {j}: words differ"
        assert abs(gw[1] - ww[1]) < 2e-4 and abs(gw[2] - ww[2]) < 2e-4
        assert gw[3] == ww[3], f"entry {j}: alignment differs"
    assert len({w for w, _, _, _ in got}) == len(got), "word sequences must be distinct"


@pytest.mark.parametrize("seed,n,fan,nbest,beam", [(0, 60, 3, 10, 4.0), (1, 200, 4, 30, 6.0), (2, 400, 3, 100, 8.0), (3, 120, 5, 50, 2.0), (4, 30, 2, 5, 50.0)])
def test_random_lattices_against_the_oracle(seed, n, fan, nbest, beam):
    rs = np.random.RandomState(seed)
    src, dst, il, ol, gr, ac = [], [], [], [], [], []
    for s in range(n - 1):
        for _ in range(fan):
            d = rs.randint(s + 1, min(n, s + 6))
            src.append(s); dst.append(d); il.append(int(rs.randint(0, 6))); ol.append(int(rs.choice([0, 0, 0, 7, 8, 9, 10])))
            gr.append(float(rs.rand())); ac.append(float(rs.rand() * 2))
    perm = rs.permutation(n)            # state ids in no particular order (the real lattices are not topologically numbered)
    src, dst = perm[np.array(src)], perm[np.array(dst)]
    fs = perm[np.array([n - 1, n - 2, n - 3])]; fc = np.array([0.25, 1.0, 0.0], np.float32)
    args = (n, int(perm[0]), src, dst, il, ol, gr, ac, fs, fc, nbest, beam)
    compare(host_nbest(*args), oracle_nbest(*args))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_negative_acoustic_costs(seed):
    """Acoustic costs -(logp - log_prior) can be negative after the DecodeNumpy prologue (lm_decoder.cc:30-35): the backward
    bound must use the same unclamped costs as the forward search, or valid paths are pruned (lattice.cpp:62)."""
    rs = np.random.RandomState(seed)
    n, fan = 150, 4
    src, dst, il, ol, gr, ac = [], [], [], [], [], []
    for s in range(n - 1):
        for _ in range(fan):
            d = rs.randint(s + 1, min(n, s + 6))
            src.append(s); dst.append(d); il.append(int(rs.randint(0, 6))); ol.append(int(rs.choice([0, 0, 0, 7, 8, 9, 10])))
            gr.append(float(rs.rand())); ac.append(float(rs.rand() * 3 - 2.0))          # mostly negative
    fs = np.array([n - 1, n - 2]); fc = np.array([0.0, -0.5], np.float32)
    args = (n, 0, np.array(src), np.array(dst), il, ol, gr, ac, fs, fc, 40, 3.0)
    compare(host_nbest(*args), oracle_nbest(*args))


def test_real_lattice_against_the_oracle():
    Z = np.load(os.path.join(ROOT, "tests", "golden", "wfst_lattice_u2.npz"))
    n_states, n_arcs, n_final, start, frames = (int(v) for v in Z["meta"])
    assert n_arcs == len(Z["src"]) and n_final == len(Z["fs"])
    args = (n_states, start, Z["src"], Z["dst"], Z["il"], Z["ol"], Z["gr"], Z["ac"], Z["fs"], Z["fc"], 100, 8.0)
    got = host_nbest(*args, cap=100 * (2 * frames + 16) + 16)
    want = oracle_nbest(*args)
    assert len(got) == 100
    compare(got, want)


def test_unreachable_and_single_state():
    # no final state reachable -> no result; a start state that is final -> the empty sequence
    assert host_nbest(3, 0, [0], [1], [2], [7], [0.5], [0.5], [2], [0.0], 5, 8.0) == []
    got = host_nbest(1, 0, [], [], [], [], [], [], [0], [1.5], 5, 8.0)
    assert got == [((), 1.5, 0.0, ())]


# ---- BrainSpeechDecoder::Rescore as lattice composition (b2t_lattice_rescore_nbest_host, csrc/graphc.cpp) --------------------------
def host_rescore(n_states, start, src, dst, il, ol, gr, ac, fs, fc, g_old, g_new, backoff, nbest, beam, cap=1 << 16):
    lib = N.load()
    A = [np.ascontiguousarray(x, dtype=t) for x, t in ((src, np.int32), (dst, np.int32), (il, np.int32), (ol, np.int32), (gr, np.float32), (ac, np.float32))]
    fs, fc = np.ascontiguousarray(fs, np.int32), np.ascontiguousarray(fc, np.float32)
    ow, oa = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
    woff, aoff, costs = np.zeros(nbest + 1, np.int32), np.zeros(nbest + 1, np.int32), np.zeros(2 * nbest, np.float32)
    stats = np.zeros(4, np.int64)
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    k = lib.b2t_lattice_rescore_nbest_host(n_states, start, len(A[0]), P(A[0]), P(A[1]), P(A[2]), P(A[3]), P(A[4]), P(A[5]), len(fs), P(fs), P(fc),
                                           g_old._h, g_new._h, backoff, nbest, C.c_float(beam), P(ow), P(woff), cap, P(oa), P(aoff), cap, P(costs),
                                           P(stats))
    assert k >= 0, N.last_error()
    return [(tuple(ow[woff[j]:woff[j + 1]].tolist()), float(costs[2 * j]), float(costs[2 * j + 1]), tuple(oa[aoff[j]:aoff[j + 1]].tolist()))
            for j in range(k)], stats


def _grammar_lists(G):
    """wfst.Fst -> (arcs[s] = [(ilabel, weight, next)], finals, start) for the oracle's grammar_min_cost"""
    arcs = [[] for _ in range(G.n)]
    for s, il, ol, w, d in G.arcs:
        arcs[s].append((il, float(w), d))
    return arcs, {int(s): float(c) for s, c in G.final.items()}, G.start


def reference_rescore(n_states, start, src, dst, il, ol, gr, ac, fs, fc, G_old, G_new, backoff, nbest, beam):
    """oracle/wfst_oracle.py: rescore_by_definition (brain_speech_decoder.cc:47-101 by enumeration)."""
    arcs = [[] for _ in range(n_states)]
    for i in range(len(src)):
        arcs[int(src[i])].append((int(il[i]), int(ol[i]), float(np.float32(gr[i])), float(np.float32(ac[i])), int(dst[i])))
    finals = {}
    for s, c in zip(fs, fc):
        finals[int(s)] = min(finals.get(int(s), float("inf")), float(np.float32(c)))
    return O.rescore_by_definition(arcs, finals, start, _grammar_lists(G_old), _grammar_lists(G_new), backoff, nbest, beam)


@pytest.mark.parametrize("seed,n,nbest,beam", [(0, 24, 10, 3.0), (1, 28, 25, 5.0), (2, 30, 200, 100.0), (3, 26, 5, 1.5), (4, 22, 40, 8.0)])
def test_rescore_by_lattice_composition_against_enumeration(seed, n, nbest, beam):
    import ngram_lm
    import wfst
    rs = np.random.RandomState(100 + seed)
    vocab = [f"w{k}" for k in range(6)]
    table = ["<eps>"] + vocab + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= len(vocab)}
    wd0 = table.index("#0")
    # two back-off grammars of different order over the same words: the graph's (2-gram) and the rescoring one (3-gram)
    G_old = wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 2, 14, seed=seed), word_id, wd0)
    G_new = wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 3, 30, seed=50 + seed), word_id, wd0)
    H_old, H_new = wfst.HostFst.from_fst(G_old).arcsort(), wfst.HostFst.from_fst(G_new).arcsort()
    src, dst, il, ol, gr, ac = [], [], [], [], [], []
    for s in range(n - 1):
        for _ in range(2):
            d = rs.randint(s + 1, min(n, s + 4))
            src.append(s); dst.append(d); il.append(int(rs.randint(0, 6))); ol.append(int(rs.choice([0, 0, 1, 2, 3, 4, 5, 6])))
            gr.append(float(rs.rand() * 2)); ac.append(float(rs.rand() * 3 - 1.0))
    perm = rs.permutation(n)
    src, dst = perm[np.array(src)], perm[np.array(dst)]
    fs = perm[np.array([n - 1, n - 2])]; fc = np.array([0.3, 0.0], np.float32)
    args = (n, int(perm[0]), src, dst, il, ol, gr, ac, fs, fc)
    got, stats = host_rescore(*args, H_old, H_new, wd0, nbest, beam)
    want = reference_rescore(*args, G_old, G_new, wd0, nbest, beam)
    assert len(want) > 0 and len(got) == len(want), (len(got), len(want))
    tot_g = np.array([g + a for _, g, a, _ in got]); tot_w = np.array([g + a for _, g, a, _ in want])
    np.testing.assert_allclose(tot_g, tot_w, rtol=0, atol=3e-4)
    n_checked = 0
    for j, (gw, ww) in enumerate(zip(got, want)):
        tied = (j > 0 and abs(tot_w[j] - tot_w[j - 1]) < 1e-3) or (j + 1 < len(want) and abs(tot_w[j + 1] - tot_w[j]) < 1e-3)
        if tied:
            continue
        assert gw[0] == ww[0], f"entry {j}: words differ"
        assert abs(gw[1] - ww[1]) < 3e-4 and abs(gw[2] - ww[2]) < 3e-4
        assert tuple(gw[3]) == tuple(ww[3]), f"entry {j}: alignment (input labels of the word sequence's best path) differs"
        n_checked += 1
    assert n_checked >= min(3, len(want)) and len({w for w, _, _, _ in got}) == len(got)
    assert stats[0] >= n // 2 and stats[2] >= 2 and stats[3] >= 2            # it did build a product with several grammar states
    # the exchange matters on this lattice: the rescored order is not the first-pass order
    first = host_nbest(*args, max(nbest, 50), beam)
    if len(want) >= 3:
        assert [w for w, _, _, _ in first[:len(got)]] != [w for w, _, _, _ in got] or seed == 3


def test_rescore_real_lattice_promotes_from_below_the_list():
    """On a real lattice (one utterance of the bench workload, 9329 arcs) with a rescoring grammar that strongly prefers some
    words: the composition ranks over ALL sequences of the lattice, so it must (a) contain, for every entry of a very deep
    first-pass list, the exchanged score the list-based exchange computes, in the right place, and (b) equal the list-based
    exchange when that list is deep enough to hold everything within the beam."""
    import ngram_lm
    import wfst
    Z = np.load(os.path.join(ROOT, "tests", "golden", "wfst_lattice_u2.npz"))
    n_states, n_arcs, n_final, start, frames = (int(v) for v in Z["meta"])
    words_used = sorted(set(int(w) for w in Z["ol"] if w > 0))
    V = max(words_used)
    vocab = [f"w{k}" for k in range(1, V + 1)]
    table = ["<eps>"] + vocab + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= V}
    wd0 = V + 1
    G_old = wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 2, 3 * V, seed=5), word_id, wd0)
    G_new = wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 3, 6 * V, seed=6), word_id, wd0)
    H_old, H_new = wfst.HostFst.from_fst(G_old).arcsort(), wfst.HostFst.from_fst(G_new).arcsort()
    args = (n_states, start, Z["src"], Z["dst"], Z["il"], Z["ol"], Z["gr"], Z["ac"], Z["fs"], Z["fc"])
    cap = 4000 * (2 * frames + 16)
    got, stats = host_rescore(*args, H_old, H_new, wd0, 50, 8.0, cap=cap)
    deep = host_nbest(*args, 4000, 8.0, cap=cap)            # everything the beam holds, if fewer than 4000
    ex = []
    for w, g, a, ali in deep:
        go, gn = H_old.grammar_score(list(w), wd0), H_new.grammar_score(list(w), wd0)
        if np.isfinite(go) and np.isfinite(gn):
            ex.append((w, g - go + gn, a))
    ex.sort(key=lambda e: e[1] + e[2])
    assert len(got) == 50
    if len(deep) < 4000:                                      # the list was exhaustive: the two must agree entry by entry
        for j in range(50):
            assert abs((got[j][1] + got[j][2]) - (ex[j][1] + ex[j][2])) < 1e-3, j
    else:                                                     # never worse than the list's exchange, entry by entry
        for j in range(50):
            assert got[j][1] + got[j][2] <= ex[j][1] + ex[j][2] + 1e-3, j
    by_words = {w: (g, a) for w, g, a in ex}
    hit = 0
    for w, g, a, _ in got:
        if w in by_words:
            assert abs(g - by_words[w][0]) < 1e-3 and abs(a - by_words[w][1]) < 1e-3
            hit += 1
    assert hit >= 25
    # a shallow list (the round-2 approximation at its old default depth relative to n) misses promoted sequences or not --
    # either way the composition's k-th total is never above the shallow exchange's k-th total
    shallow = sorted((g - H_old.grammar_score(list(w), wd0) + H_new.grammar_score(list(w), wd0) + a) for w, g, a, _ in deep[:60])
    assert all(got[j][1] + got[j][2] <= shallow[j] + 1e-3 for j in range(min(50, len(shallow))))


def test_rescore_on_the_determinised_lattice_equals_the_raw_composition(monkeypatch):
    """Round 4: Rescore() determinises the lattice over words before composing it with the grammars (as the reference composes
    lat_, lattice-faster-decoder.cc:193-213 + brain_speech_decoder.cc:47-101).  On a real lattice: the same word sequences in the
    same order with the same graph / acoustic costs as the composition of the RAW lattice (B2T_RESCORE_RAW=1, the round-3 form),
    a product two orders of magnitude smaller, and alignments that span every frame."""
    import ngram_lm
    import wfst
    Z = np.load(os.path.join(ROOT, "tests", "golden", "wfst_lattice_u2.npz"))
    n_states, n_arcs, n_final, start, frames = (int(v) for v in Z["meta"])
    V = max(int(w) for w in Z["ol"] if w > 0)
    vocab = [f"w{k}" for k in range(1, V + 1)]
    table = ["<eps>"] + vocab + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= V}
    wd0 = V + 1
    H_old = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 2, 3 * V, seed=5), word_id, wd0)).arcsort()
    H_new = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 3, 6 * V, seed=6), word_id, wd0)).arcsort()
    args = (n_states, start, Z["src"], Z["dst"], Z["il"], Z["ol"], Z["gr"], Z["ac"], Z["fs"], Z["fc"])
    cap = 400 * (2 * frames + 16)
    monkeypatch.setenv("B2T_RESCORE_RAW", "1")
    raw, st_raw = host_rescore(*args, H_old, H_new, wd0, 200, 8.0, cap=cap)
    monkeypatch.setenv("B2T_RESCORE_RAW", "0")
    det, st_det = host_rescore(*args, H_old, H_new, wd0, 200, 8.0, cap=cap)
    assert len(det) == len(raw) >= 50
    tr, td = np.array([g + a for _, g, a, _ in raw]), np.array([g + a for _, g, a, _ in det])
    np.testing.assert_allclose(td, tr, atol=2e-3)
    n_same_ali = n_untied = 0
    for j, (x, y) in enumerate(zip(det, raw)):
        tied = (j > 0 and abs(tr[j] - tr[j - 1]) < 2e-3) or (j + 1 < len(raw) and abs(tr[j + 1] - tr[j]) < 2e-3)
        if not tied:
            assert x[0] == y[0], j
            assert abs(x[1] - y[1]) < 2e-3 and abs(x[2] - y[2]) < 2e-3, j
            n_same_ali += x[3] == y[3]                  # (round 5: the determinised path recomputes back pointers on demand, DetRescore::trace)
            n_untied += 1
        assert len(x[3]) == frames == len(y[3]), "an alignment has one input label per decoded frame"
    assert det[0][3] == raw[0][3]                       # the best hypothesis' alignment (no tie at the top of this lattice)
    print(f"alignments equal to the raw composition's: {n_same_ali} of {n_untied} untied hypotheses")
    assert n_same_ali == n_untied                       # every untied hypothesis: the alignment of its best path, label for label
    assert {w for w, *_ in det} == {w for w, *_ in raw} and len({w for w, *_ in det}) == len(det)
    assert st_det[1] * 10 < st_raw[1], (st_det, st_raw)  # product arcs: determinised first vs raw


def test_rescore_work_arrays_are_reused_without_leaking_between_calls():
    """Round 5: the determinisation's work arrays live per host thread and are reused from call to call (csrc/graphc.cpp,
    DetRescore).  Lattices of different sizes alternating on one thread, and the same mix on several threads at once (the
    pool of wfst_decoder._nbest_host), must each return exactly what a first call returned: words, alignments, float costs."""
    from concurrent.futures import ThreadPoolExecutor
    import ngram_lm
    import wfst
    Z = np.load(os.path.join(ROOT, "tests", "golden", "wfst_lattice_u2.npz"))
    n_states, n_arcs, n_final, start, frames = (int(v) for v in Z["meta"])
    V = max(int(w) for w in Z["ol"] if w > 0)
    vocab = [f"w{k}" for k in range(1, V + 1)]
    table = ["<eps>"] + vocab + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= V}
    wd0 = V + 1
    H_old = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 2, 3 * V, seed=5), word_id, wd0)).arcsort()
    H_new = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 3, 6 * V, seed=6), word_id, wd0)).arcsort()
    real = (n_states, start, Z["src"], Z["dst"], Z["il"], Z["ol"], Z["gr"], Z["ac"], Z["fs"], Z["fc"])
    cases = [(real, 100, 8.0, 200 * (2 * frames + 16))]
    for seed in range(3):                                  # small random lattices over the same word ids (ragged against the real one)
        rs = np.random.RandomState(700 + seed)
        n = 20 + 7 * seed
        src, dst, il, ol, gr, ac = [], [], [], [], [], []
        for s in range(n - 1):
            for _ in range(2):
                src.append(s); dst.append(int(rs.randint(s + 1, min(n, s + 4)))); il.append(int(rs.randint(0, 6)))
                ol.append(int(rs.choice([0, 0, 1, 2, 3, min(4, V)]))); gr.append(float(rs.rand() * 2)); ac.append(float(rs.rand() * 3 - 1.0))
        cases.append(((n, 0, src, dst, il, ol, gr, ac, [n - 1, n - 2], [0.3, 0.0]), 20 + 10 * seed, 4.0 + seed, 1 << 16))
    run = lambda k: host_rescore(*cases[k][0], H_old, H_new, wd0, cases[k][1], cases[k][2], cap=cases[k][3])[0]
    first = [run(k) for k in range(len(cases))]
    assert all(len(f) > 0 for f in first) and len(first[0]) >= 50
    for k in (0, 1, 0, 3, 2, 0, 1):                        # big, small, big, ...: one thread's arrays shrink and grow logically
        assert run(k) == first[k], k
    order = [0, 1, 2, 3] * 6
    with ThreadPoolExecutor(max_workers=4) as pool:
        for k, got in zip(order, pool.map(run, order)):
            assert got == first[k], k


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("seed", range(12))
def test_rescore_epsilon_heavy_lattices_against_enumeration(seed, threads, monkeypatch):
    """Round 5: lattices that are mostly epsilon-output arcs (25-80 %), fan-out 1-3: chains, and single-exit states with SEVERAL
    incoming arcs, which the determinisation now absorbs into each of them (csrc/graphc.cpp, DetRescore::setup) -- words, graph /
    acoustic costs and the alignment of every untied entry against the enumeration (oracle/wfst_oracle.py rescore_by_definition).
    (600 further seeds of this generator were run once when the code was written: 0 differences.)"""
    import ngram_lm
    import wfst
    # threads = 3: states are expanded ahead of their turn by helper threads, in batches of >= 2 here (192 in production, where
    # only lattices of >= 60 k arcs get helpers): the answers must not know
    monkeypatch.setenv("B2T_RESCORE_THREADS", str(threads)); monkeypatch.setenv("B2T_RESCORE_MIN_BATCH", "2")
    rs = np.random.RandomState(9000 + seed)
    n = int(rs.randint(8, 34)); nbest = int(rs.choice([5, 20, 60, 200])); beam = float(rs.choice([1.5, 3.0, 6.0, 100.0]))
    vocab = [f"w{k}" for k in range(6)]
    table = ["<eps>"] + vocab + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= len(vocab)}
    wd0 = table.index("#0")
    G_old = wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 2, 14, seed=seed), word_id, wd0)
    G_new = wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 3, 30, seed=50 + seed), word_id, wd0)
    H_old, H_new = wfst.HostFst.from_fst(G_old).arcsort(), wfst.HostFst.from_fst(G_new).arcsort()
    src, dst, il, ol, gr, ac = [], [], [], [], [], []
    fan = int(rs.choice([1, 2, 2, 3]))
    p_eps = float(rs.choice([0.25, 0.5, 0.8]))
    for s in range(n - 1):
        for _ in range(fan if rs.rand() < 0.7 else 1):
            d = rs.randint(s + 1, min(n, s + 4))
            src.append(s); dst.append(d); il.append(int(rs.randint(0, 6))); ol.append(0 if rs.rand() < p_eps else int(rs.randint(1, 7)))
            gr.append(float(rs.rand() * 2)); ac.append(float(rs.rand() * 3 - 1.0))
    perm = rs.permutation(n)
    src, dst = perm[np.array(src)], perm[np.array(dst)]
    fs = perm[np.array([n - 1, n - 2])]; fc = np.array([0.3, 0.0], np.float32)
    args = (n, int(perm[0]), src, dst, il, ol, gr, ac, fs, fc)
    got, _ = host_rescore(*args, H_old, H_new, wd0, nbest, beam)
    want = reference_rescore(*args, G_old, G_new, wd0, nbest, beam)
    assert len(got) == len(want)
    tot_w = np.array([g + a for _, g, a, _ in want]); tot_g = np.array([g + a for _, g, a, _ in got])
    if len(want):
        np.testing.assert_allclose(tot_g, tot_w, rtol=0, atol=3e-4)
    for j, (gw, ww) in enumerate(zip(got, want)):
        tied = (j > 0 and abs(tot_w[j] - tot_w[j - 1]) < 1e-3) or (j + 1 < len(want) and abs(tot_w[j + 1] - tot_w[j]) < 1e-3)
        if tied:
            continue
        assert gw[0] == tuple(ww[0]) and abs(gw[1] - ww[1]) < 3e-4 and abs(gw[2] - ww[2]) < 3e-4, j
        assert tuple(gw[3]) == tuple(ww[3]), j


def test_rescore_with_helper_threads_is_bit_identical_on_the_real_lattice(monkeypatch):
    """Round 5: DetRescore::run with n_threads > 1 prepares the expansions of queued states in parallel batches and redoes the
    few whose forward cost was lowered past a refused candidate -- words, alignments and float costs must equal the serial
    determinisation's bit for bit, whatever the thread count and batch threshold."""
    import ngram_lm
    import wfst
    Z = np.load(os.path.join(ROOT, "tests", "golden", "wfst_lattice_u2.npz"))
    n_states, n_arcs, n_final, start, frames = (int(v) for v in Z["meta"])
    V = max(int(w) for w in Z["ol"] if w > 0)
    vocab = [f"w{k}" for k in range(1, V + 1)]
    table = ["<eps>"] + vocab + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= V}
    wd0 = V + 1
    H_old = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 2, 3 * V, seed=5), word_id, wd0)).arcsort()
    H_new = wfst.HostFst.from_fst(wfst.grammar_fst(ngram_lm.synthetic_word_arpa(vocab, 3, 6 * V, seed=6), word_id, wd0)).arcsort()
    args = (n_states, start, Z["src"], Z["dst"], Z["il"], Z["ol"], Z["gr"], Z["ac"], Z["fs"], Z["fc"])
    cap = 400 * (2 * frames + 16)
    monkeypatch.setenv("B2T_RESCORE_THREADS", "1")
    serial, _ = host_rescore(*args, H_old, H_new, wd0, 200, 8.0, cap=cap)
    assert len(serial) >= 50
    for threads, min_batch in ((2, 2), (4, 16), (3, 192), (6, 64)):
        monkeypatch.setenv("B2T_RESCORE_THREADS", str(threads)); monkeypatch.setenv("B2T_RESCORE_MIN_BATCH", str(min_batch))
        for _ in range(2):
            got, _ = host_rescore(*args, H_old, H_new, wd0, 200, 8.0, cap=cap)
            assert got == serial, (threads, min_batch)
