"""Decode-graph construction (nejm-brain-to-text_amd/wfst.py) and the oracle's WFST decoder (oracle/wfst_oracle.py), CPU only.

The reference pins nothing on this path (no graph, no logits fixture, no expected n-best in the checkout; its C++ cannot be
built here): the checks below are structural facts of the recipes the modules restate, plus independent computations of
the same quantities (a dense Viterbi over the graph for the best path, exhaustive path enumeration for the n-best)."""
import itertools
import math
import os

import numpy as np
import pytest

import ngram_lm
import wfst
from oracle import wfst_oracle as W

ARPA = """\\data\\
ngram 1=5
ngram 2=4

\\1-grams:
-1.0 <s> -0.5
-0.7 </s>
-0.4 a -0.3
-0.6 b -0.2
-2.0 <unk> -0.1

\\2-grams:
-0.2 <s> a
-0.3 a b
-0.5 b </s>
-0.9 a </s>

\\end\\
"""
LN10 = math.log(10.0)


def test_token_fst_is_the_corrected_ctc_topology():
    """tools/fst/ctc_token_fst_corrected.py:42-57: 1 blank loop, 3 arcs per unit, n(n-1) unit-to-unit arcs, one #k self-loop
    per state and disambiguation symbol; every state final; blank ilabel 1, unit i ilabel = olabel = i + 1."""
    n, dis = 5, [7, 8]
    t = wfst.token_fst(n, dis)
    assert t.n == n + 1 and t.start == 0 and set(t.final) == set(range(n + 1))
    assert len(t.arcs) == 1 + 3 * n + n * (n - 1) + (n + 1) * len(dis)
    assert (0, 1, 0, 0.0, 0) in t.arcs
    for i in range(1, n + 1):
        assert (0, i + 1, i + 1, 0.0, i) in t.arcs and (i, i + 1, 0, 0.0, i) in t.arcs and (i, 1, 0, 0.0, 0) in t.arcs
    assert (2, 4, 4, 0.0, 3) in t.arcs and not any(a[0] == a[4] and a[2] == a[1] and a[1] > 1 and a[1] < 7 for a in t.arcs)
    assert sum(1 for a in t.arcs if a[1] == 0) == (n + 1) * len(dis)


def test_grammar_fst_follows_arpa2fst_with_the_recipe_s_relabelling():
    """arpa-lm-compiler.cc:162-285 + eps2disambig.pl + s2eps.pl + fstrmepsilon (make_tlg.sh:29-40): start = the <s>
    history, one state per history, highest-order n-grams go to the back-off history, </s> becomes a final cost, back-off
    arcs carry #0 on the input side, <unk> lines are dropped."""
    word_id = {"a": 1, "b": 2}
    g = wfst.grammar_fst(ARPA, word_id, 3)
    out = g.out()
    arcs = {(s, il): (w, d) for s in range(g.n) for il, ol, w, d in out[s]}
    s_bos = g.start
    w, s_a = arcs[(s_bos, 1)]                    # "<s> a" (highest order) -> history (a)
    assert abs(w - 0.2 * LN10) < 1e-6
    w, s_uni = arcs[(s_bos, 3)]                  # back-off of <s> to the zerogram state
    assert abs(w - 0.5 * LN10) < 1e-6
    w, s_b = arcs[(s_a, 2)]                      # "a b"
    assert abs(w - 0.3 * LN10) < 1e-6
    assert abs(g.final[s_a] - 0.9 * LN10) < 1e-6 and abs(g.final[s_b] - 0.5 * LN10) < 1e-6 and abs(g.final[s_uni] - 0.7 * LN10) < 1e-6
    assert abs(arcs[(s_uni, 1)][0] - 0.4 * LN10) < 1e-6 and arcs[(s_uni, 1)][1] == s_a
    assert abs(arcs[(s_a, 3)][0] - 0.3 * LN10) < 1e-6 and arcs[(s_a, 3)][1] == s_uni
    assert all(ol == 0 for s in range(g.n) for il, ol, w, d in out[s] if il == 3)
    # the LM cost of a sentence = the cheapest path (n-gram or back-off), e.g. "a b": -log p(a|<s>) p(b|a) p(</s>|b)
    assert abs(wfst.grammar_score(g, [1, 2], 3) - (0.2 + 0.3 + 0.5) * LN10) < 1e-6
    assert abs(wfst.grammar_score(g, [2], 3) - (0.5 + 0.6 + 0.5) * LN10) < 1e-6      # <s> backs off, then unigram b
    assert wfst.grammar_score(g, [5], 3) == math.inf


def _paths(f, max_len):
    """{(istring, ostring): min cost} over all accepting paths of at most max_len arcs."""
    out, res = f.out(), {}
    stack = [(f.start, (), (), 0.0, 0)]
    while stack:
        s, i, o, c, n = stack.pop()
        if s in f.final:
            k = (i, o)
            res[k] = min(res.get(k, math.inf), c + f.final[s])
        if n < max_len:
            for il, ol, w, d in out[s]:
                stack.append((d, i + ((il,) if il else ()), o + ((ol,) if ol else ()), c + w, n + 1))
    return res


def test_compose_equals_relational_composition():
    """Epsilon-filtered composition: for every input string x and output string z, cost(x, z) = min over y of
    A(x, y) + B(y, z) -- checked by exhaustive path enumeration on random small transducers with epsilons on both sides."""
    rs = np.random.RandomState(3)
    for trial in range(4):
        def rnd(n_states, n_arcs, labels_in, labels_out):
            f = wfst.Fst()
            for _ in range(n_states):
                f.add_state()
            f.start = 0
            for _ in range(n_arcs):
                s = rs.randint(n_states); d = rs.randint(s, n_states)      # forward or self arcs
                if s == d and rs.rand() < 0.7:
                    d = min(n_states - 1, s + 1)
                f.add_arc(s, int(rs.choice(labels_in)), int(rs.choice(labels_out)), float(np.round(rs.rand() * 3, 2)), d)
            f.final[n_states - 1] = float(np.round(rs.rand(), 2))
            return f
        A = rnd(4, 9, [0, 1, 2], [0, 5, 6])
        B = rnd(4, 9, [0, 5, 6], [0, 8, 9])
        Cf = wfst.compose(A, B)
        got = _paths(Cf, 7)
        pa, pb = _paths(A, 5), _paths(B, 5)
        want = {}
        for (x, y), ca in pa.items():
            for (y2, z), cb in pb.items():
                if y == y2:
                    want[(x, z)] = min(want.get((x, z), math.inf), ca + cb)
        # A and B are acyclic apart from self loops; restrict to pairs both enumerations can see completely
        for k, c in want.items():
            if len(k[0]) <= 2 and len(k[1]) <= 2 and k in got:
                assert abs(got[k] - c) < 1e-6, (trial, k, got[k], c)
        assert set(k for k in want if len(k[0]) + len(k[1]) <= 2) <= set(got)


@pytest.fixture(scope="module")
def toy():
    prons = ngram_lm.synthetic_lexicon(30, 41, seed=1)
    words = sorted(prons)
    arpa = ngram_lm.synthetic_word_arpa(words, 3, 150, seed=2)
    return prons, words, wfst.build_tlg(prons, arpa, sil_prob=0.5)


def _spell(prons, seq, rs, noise=0.5, blank_bias=0.0):
    frames = []
    for w in seq:
        for c in list(prons[w][0]) + [1]:
            frames += [c, 0]
    lg = np.full((len(frames), 41), -3.0, np.float32)
    for t, c in enumerate(frames):
        lg[t, c] = 4.0
    lg += rs.standard_normal(lg.shape).astype(np.float32) * noise
    lp = lg - np.log(np.exp(lg).sum(-1, keepdims=True))
    lp[:, 0] -= blank_bias
    return lp.astype(np.float32)


def test_graph_decodes_spelled_sentences(toy):
    """Peaked log-probabilities that spell a word sequence (SIL after every word) decode to that sequence; the ilabels of
    the graph are the decoder classes + 1 (blank 1, SIL 2, phonemes 3..41) and no disambiguation symbol survives."""
    prons, words, g = toy
    assert g.ilabel.max() == 41 and g.ilabel.min() == 0 and (g.n_eps > 0).any() and np.isfinite(g.final).any()
    rs = np.random.RandomState(0)
    cfg = W.Config(beam=17, max_active=7000, min_active=200, lattice_beam=8, acoustic_scale=0.325, nbest=5, blank_skip_thresh=1.0)
    for trial in range(3):
        seq = [words[i] for i in rs.randint(len(words), size=3)]
        S = W.CtcWfstBeamSearch(g, cfg)
        S.search(_spell(prons, seq, rs))
        assert [g.words[w] for w in S.outputs[0]] == seq
        S.finalize_search()
        res = W.decode_results(S, g.words)
        assert res[0][0] == " ".join(seq) and len(res) >= 1
        assert all(res[i][1] * 0.325 + res[i][2] >= res[i + 1][1] * 0.325 + res[i + 1][2] - 1e-4 for i in range(len(res) - 1))
        # the phoneme string of the best entry is the pronunciation (with SILs), its times increase
        want = [c for w in seq for c in list(prons[w][0]) + [1]]
        assert S.inputs[0] == want and all(a < b for a, b in zip(S.times[0], S.times[0][1:]))


def test_oracle_best_path_equals_dense_viterbi(toy):
    """With pruning out of the way (huge beams) the token-passing decoder's best cost is the Viterbi cost over the graph: an
    independent dense DP (min-plus over all arcs per frame + epsilon closure by relaxation)."""
    prons, words, g = toy
    rs = np.random.RandomState(5)
    T = 14
    lp = np.log(rs.dirichlet(np.ones(41) * 0.3, size=T)).astype(np.float32)
    cfg = W.Config(beam=1e4, max_active=2 ** 31 - 1, min_active=0, lattice_beam=1e4, acoustic_scale=0.5, nbest=1, blank_skip_thresh=1.0)
    S = W.CtcWfstBeamSearch(g, cfg)
    S.search(lp)
    S.finalize_search()
    lm, ac = S.likelihood[0]
    src = np.repeat(np.arange(g.n_states), np.diff(g.row))
    eps = g.ilabel == 0

    def closure(c):
        for _ in range(50):
            cand = c[src[eps]] + g.weight[eps]
            new = c.copy()
            np.minimum.at(new, g.next[eps], cand)
            if np.array_equal(new, c):
                break
            c = new
        return c
    cost = np.full(g.n_states, np.inf); cost[g.start] = 0.0
    cost = closure(cost)
    em = ~eps
    for t in range(T):
        cand = cost[src[em]] + g.weight[em] - 0.5 * lp[t, g.ilabel[em] - 1]
        new = np.full(g.n_states, np.inf)
        np.minimum.at(new, g.next[em], cand)
        cost = closure(new)
    best = float(np.min(cost + g.final))
    assert abs(-(lm + ac) - best) < 1e-3 * max(1.0, abs(best))


def test_oracle_nbest_equals_exhaustive_enumeration():
    """The n-best of a lattice = the cheapest path of every distinct word sequence, sorted: checked against brute-force
    enumeration of all paths of a small decode (2-word lexicon, 6 frames)."""
    prons = {"ab": [(5, 9)], "ba": [(9, 5)], "a": [(5,)]}
    arpa = ngram_lm.synthetic_word_arpa(sorted(prons), 2, 6, seed=4)
    g = wfst.build_tlg(prons, arpa, sil_prob=0.3)
    rs = np.random.RandomState(2)
    lp = np.log(rs.dirichlet(np.ones(41), size=6)).astype(np.float32)
    lp[:, [0, 1, 5, 9]] += 3.0
    cfg = W.Config(beam=30, max_active=7000, min_active=200, lattice_beam=6.0, acoustic_scale=0.6, nbest=50, blank_skip_thresh=1.0)
    S = W.CtcWfstBeamSearch(g, cfg)
    S.search(lp)
    S.finalize_search()
    arcs, finals, start = S.dec.raw_lattice()
    best = {}
    stack = [(start, (), 0.0)]
    while stack:
        s, w, c = stack.pop()
        if s in finals:
            best[w] = min(best.get(w, math.inf), c + finals[s])
        for il, ol, gc, ac, d in arcs[s]:
            stack.append((d, w + ((ol,) if ol else ()), c + gc + ac))
    ranked = sorted(best.items(), key=lambda kv: kv[1])
    lim = ranked[0][1] + 6.0
    ranked = [(w, c) for w, c in ranked if c <= lim + 1e-4]
    got = [(tuple(o), -(l[0] + l[1])) for o, l in zip(S.outputs, S.likelihood)]
    assert len(got) == min(50, len(ranked)) and len(got) >= 3
    for (w1, c1), (w2, c2) in zip(got, ranked):
        assert abs(c1 - c2) < 1e-4 and (w1 == w2 or abs(c1 - dict(ranked)[w1]) < 1e-4)


def test_blank_frame_skipping_and_frame_mapping(toy):
    """ctc_wfst_beam_search.cc:79-94: frames whose blank probability exceeds the threshold are not decoded; a skipped
    blank frame is re-inserted when the same best symbol repeats across it; times refer to INPUT frames."""
    prons, words, g = toy
    rs = np.random.RandomState(7)
    seq = [words[3], words[3]]
    lp = _spell(prons, seq, rs, noise=0.1)
    lp[1::2, 0] = math.log(0.995)                       # the interleaved blank frames are near-certain blanks
    cfg = W.Config(beam=17, max_active=7000, min_active=200, lattice_beam=8, acoustic_scale=0.5, nbest=3, blank_skip_thresh=0.98)
    S = W.CtcWfstBeamSearch(g, cfg)
    S.search(lp)
    assert S.num_frames == lp.shape[0] and len(S.mapping) < lp.shape[0]
    assert all(m % 2 == 0 for m in S.mapping) or any(m % 2 == 1 for m in S.mapping)
    S.finalize_search()
    assert [g.words[w] for w in S.outputs[0]] == seq
    assert all(t in S.mapping for t in S.times[0])
    # chunked search == one shot
    S2 = W.CtcWfstBeamSearch(g, cfg)
    S2.search(lp[:7]); S2.search(lp[7:])
    S2.finalize_search()
    assert S2.outputs == S.outputs and S2.mapping == S.mapping
    np.testing.assert_allclose(S2.likelihood, S.likelihood, rtol=1e-6)


def test_openfst_vector_container_round_trip(toy, tmp_path):
    prons, words, g = toy
    f = wfst.Fst()
    f.n, f.start = g.n_states, g.start
    src = np.repeat(np.arange(g.n_states), np.diff(g.row))
    f.arcs = [(int(s), int(il), int(ol), float(w), int(d)) for s, il, ol, w, d in zip(src, g.ilabel, g.olabel, g.weight, g.next)]
    f.final = {int(s): float(c) for s, c in enumerate(g.final) if np.isfinite(c)}
    p = str(tmp_path / "TLG.fst")
    wfst.write_openfst_vector(f, p)
    with open(str(tmp_path / "words.txt"), "w") as fh:
        for i, w in enumerate(g.words):
            fh.write(f"{w} {i}\n")
    g2 = wfst.graph_from_files(p, str(tmp_path / "words.txt"))
    for k in ("row", "ilabel", "olabel", "weight", "next", "n_eps", "final"):
        np.testing.assert_array_equal(getattr(g2, k), getattr(g, k), err_msg=k)
    assert g2.words == g.words and g2.start == g.start
    wfst.save_graph(g, str(tmp_path / "tlg.npz"))
    g3 = wfst.graph_from_files(str(tmp_path / "tlg.npz"), "")
    np.testing.assert_array_equal(g3.next, g.next)
    assert g3.words == g.words
    with open(p, "r+b") as fh:
        fh.write(b"\0\0\0\0")
    with pytest.raises(ValueError, match="not an OpenFST"):
        wfst.read_openfst_vector(p)


def test_convert_rows_to_inputs_equals_the_scalar_form():
    """wfst_decoder.convert_rows_to_inputs (all utterances' partial best paths in one vectorised pass, the per-frame host side of a
    streaming decoder) == CtcWfstBeamSearch::ConvertToInputs (ctc_wfst_beam_search.cc:162-188) row by row."""
    import wfst_decoder as WD
    rs = np.random.RandomState(0)
    for _ in range(300):
        U, ML = rs.randint(1, 6), rs.randint(1, 40)
        ali = rs.choice([1, 1, 1, 2, 3, 3, 5, 7], size=(U, ML)).astype(np.int32)
        fr = np.sort(rs.randint(0, 100, size=(U, ML)), axis=1).astype(np.int32)
        n = rs.randint(0, ML + 1, size=U)
        a, b = WD.convert_rows_to_inputs(ali, fr, n)
        for u in range(U):
            i, t = WD.convert_to_inputs(ali[u, :n[u]], fr[u, :n[u]])
            assert a[u] == i and b[u] == t


def test_final_pruning_uses_kaldis_approx_equal():
    """PruneForwardLinksFinal (lattice-faster-decoder.cc:380-470) iterates while a token's extra cost is not ApproxEqual to its
    old one, and ApproxEqual (kaldi/base/kaldi-math.h:265-273) calls an infinite difference unequal.  Graph: 0 -a-> 1 (final),
    1 -eps-> 2 (dead end).  On the last frame the token of state 1 is visited first and keeps its epsilon link (the destination
    still carries extra cost 0); the dead end then goes 0 -> inf, which must count as a change so that a second sweep removes the
    link -- otherwise it outlives its destination and GetRawLattice meets a token that is gone."""
    import wfst
    f = wfst.Fst()
    for _ in range(3):
        f.add_state()
    f.start = 0
    f.add_arc(0, 2, 1, 0.5, 1)
    f.add_arc(1, 0, 0, 0.25, 2)
    f.final[1] = 0.0
    g = wfst.DecodeGraph(f, ["<eps>", "w"])
    lp = np.full((1, 41), -5.0, np.float32); lp[0, 1] = -0.1
    R = W.CtcWfstBeamSearch(g, W.Config(nbest=5))
    R.search(lp)
    R.finalize_search()
    assert [list(o) for o in R.outputs] == [[1]] and [list(i) for i in R.inputs] == [[1]]
    arcs, finals, start = R.dec.raw_lattice()
    assert sum(len(a) for a in arcs) == 1 and len(finals) == 1


def test_hash_list_hands_the_elements_out_in_the_reference_order():
    """kaldi/util/hash-list-inl.h:137-172 (Insert) / :45-58 (Clear): a key goes to bucket key % size; a bucket that becomes occupied
    is linked behind the last occupied one, a key of an occupied bucket goes behind that bucket's last element -> the list is the
    buckets in order of first occupation, each bucket in insertion order.  Worked by hand for size 5 and the insertions
    7, 3, 12, 8, 2, 13, 5:  buckets 2 (7, 12, 2), 3 (3, 8, 13), 0 (5)."""
    h = W.HashList(5)
    for k in (7, 3, 12, 8, 2, 13, 5):
        assert h.get(k) is None
        h.insert(k, f"v{k}")
    assert [k for k, _ in h.items()] == [7, 12, 2, 3, 8, 13, 5]
    assert h.get(12) == "v12" and h.get(4) is None
    assert h.clear() == [f"v{k}" for k in (7, 12, 2, 3, 8, 13, 5)]
    assert h.items() == [] and h.get(7) is None
    h.set_size(1000)                      # SetSize only on the empty list (:37-43); below 1000 keys the order is insertion order
    for k in (7, 3, 12, 8):               # per bucket, i.e. first-occupation order = insertion order when no two keys share one
        h.insert(k, k)
    assert h.values() == [7, 3, 12, 8]


def test_order_dependent_cutoff_only_matters_while_max_active_binds():
    """lattice-faster-decoder.cc:786-810: ProcessEmitting tightens next_cutoff while it walks the hash list, so tokens beyond the
    frame's final cutoff exist or not depending on list order; the next frame's GetCutoff counts them (:650-720).  The oracle's
    "sequential" rule restates that walk in HashList order; its "final" rule is the data-parallel form csrc/wfst.hip implements
    (every candidate against the frame's final cutoff).  With max_active far from binding the two give identical n-best lists;
    with max_active = 60 (binding in every frame of these graphs) they may differ in the tail of the list but not in the best
    hypothesis (tools/r5_cutoff_order.py over the 9 fuzz graphs x 21 utterances: max_active 60 -> 1 list differs, from rank 4 on;
    150 and 400 -> none)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import r5_cutoff_order as R
    rows = R.run([60, 7000], seeds=(101,), verbose=False)
    loose = [r for r in rows if r["max_active"] == 7000]
    tight = [r for r in rows if r["max_active"] == 60]
    assert loose and all(r["first_diff"] < 0 for r in loose)
    assert tight and all(r["best_same"] for r in tight)
    assert all(r["first_diff"] < 0 or r["first_diff"] >= 1 for r in tight)
