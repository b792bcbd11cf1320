"""Stage-2 (decoder) GPU parity: DecodeNumpy prologue and the batched CTC prefix beam search against the
reference's known answer (ctc_prefix_beam_search_test.cc:18-59) and the oracle restatement on random inputs,
including chunked (streaming) feeding."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import b2t_oracle as O


def _search(logp, first, second, chunks=None, max_len=None):
    import ctypes as C
    import b2t_native as N
    import b2t_ops as ops
    dev = torch.device("cuda:0")
    lib = N.load()
    U, T, Cc = logp.shape
    L = max_len or (T + 1)
    NN = T * second + 2
    state = torch.empty((lib.b2t_beam_state_bytes(L, NN) * U,), dtype=torch.uint8, device=dev)
    N.check(lib.b2t_beam_reset(ops._p(state), U, L, NN, ops._stream()), "reset")
    hyps = torch.zeros((U, second, L), dtype=torch.int32, device=dev)
    hl = torch.empty((U, second), dtype=torch.int32, device=dev)
    sc = torch.empty((U, second), device=dev); vs = torch.empty((U, second), device=dev)
    tm = torch.zeros((U, second, L), dtype=torch.int32, device=dev)
    bounds = chunks or [(0, T)]
    for a, b in bounds:
        lp = torch.from_numpy(np.ascontiguousarray(logp[:, a:b])).to(dev)
        N.check(lib.b2t_prefix_beam_search_f32(ops._p(lp), None, U, b - a, Cc, first, second, 0, ops._p(state), L, NN,
                                               ops._p(hyps), ops._p(hl), ops._p(sc), ops._p(vs), ops._p(tm), ops._stream()),
                "search")
    flag = C.c_int(0)
    N.check(lib.b2t_beam_overflowed(ops._p(state), U, L, NN, C.byref(flag), ops._stream()), "ovf")
    assert flag.value == 0
    return hyps.cpu().numpy(), hl.cpu().numpy(), sc.cpu().numpy(), vs.cpu().numpy(), tm.cpu().numpy()


def test_prefix_beam_known_answer():
    p = np.array([[0.25, 0.40, 0.35], [0.40, 0.35, 0.25], [0.10, 0.50, 0.40]], dtype=np.float32)
    hyps, hl, sc, vs, tm = _search(np.log(p)[None], 3, 3)
    got = [tuple(hyps[0, i, :hl[0, i]]) for i in range(3)]
    assert got == [(2, 1), (1, 2), (1,)]
    np.testing.assert_allclose(np.exp(sc[0]), [0.2185, 0.1550, 0.1525], rtol=1e-5)
    np.testing.assert_allclose(np.exp(vs[0]), [0.07, 0.064, 0.07], rtol=1e-5)
    assert [list(tm[0, i, :hl[0, i]]) for i in range(3)] == [[0, 2], [0, 2], [2]]


@pytest.mark.parametrize("first,second", [(10, 10), (5, 4), (16, 32)])
def test_prefix_beam_vs_oracle_random(first, second):
    rng = np.random.default_rng(first * 100 + second)
    U, T, C = 6, 60, 41
    logits = rng.standard_normal((U, T, C)).astype(np.float32) * 2.5
    logits[..., 0] += 1.0
    logp = O.log_softmax(logits)
    hyps, hl, sc, vs, tm = _search(logp, first, second)
    for u in range(U):
        ref = O.prefix_beam_search(logp[u], first, second)
        assert len(ref) == int((hl[u] >= 0).sum())
        ref_map = {tuple(r[0]): r for r in ref}
        got_scores = []
        for i in range(len(ref)):
            pre = tuple(int(v) for v in hyps[u, i, :hl[u, i]])
            assert pre in ref_map, (u, i, pre)
            r = ref_map[pre]
            assert abs(sc[u, i] - r[1]) < 2e-4 * max(1.0, abs(r[1])), (u, i)
            assert abs(vs[u, i] - r[2]) < 2e-4 * max(1.0, abs(r[2])), (u, i)
            got_scores.append(sc[u, i])
        assert all(got_scores[i] >= got_scores[i + 1] - 1e-6 for i in range(len(got_scores) - 1))   # sorted
        # best hypothesis and its Viterbi token times agree exactly
        assert tuple(int(v) for v in hyps[u, 0, :hl[u, 0]]) == tuple(ref[0][0])
        assert list(tm[u, 0, :hl[u, 0]]) == ref[0][3]


@pytest.mark.parametrize("first,second", [(10, 100), (16, 128)])
def test_prefix_beam_wide_vs_oracle(first, second):
    """BASELINE configs[3] quotes beam = 100: wide second beams (candidate arrays in dynamic LDS).  Far down the beam
    the scores are dense, so fp32-vs-oracle ties at the pruning boundary may swap a few prefixes: the upper half must
    agree exactly (prefix, score, order), the whole beam to >= 95 %."""
    rng = np.random.default_rng(first + second)
    U, T, C = 3, 40, 41
    logits = rng.standard_normal((U, T, C)).astype(np.float32) * 1.5
    logits[..., 0] += 1.0
    logp = O.log_softmax(logits)
    hyps, hl, sc, vs, tm = _search(logp, first, second)
    for u in range(U):
        ref = O.prefix_beam_search(logp[u], first, second)
        n = int((hl[u] >= 0).sum())
        assert n == len(ref) == second
        got = [tuple(int(v) for v in hyps[u, i, :hl[u, i]]) for i in range(n)]
        ref_map = {tuple(r[0]): r for r in ref}
        for i in range(second // 2):
            assert got[i] == tuple(ref[i][0]), (u, i)
            assert abs(sc[u, i] - ref[i][1]) < 2e-4 * max(1.0, abs(ref[i][1]))
            assert abs(vs[u, i] - ref[i][2]) < 2e-4 * max(1.0, abs(ref[i][2]))
        assert sum(g in ref_map for g in got) >= int(0.95 * second)
        assert all(sc[u, i] >= sc[u, i + 1] - 1e-6 for i in range(n - 1))
        assert list(tm[u, 0, :hl[u, 0]]) == ref[0][3]


def test_prefix_beam_streaming_equals_offline():
    rng = np.random.default_rng(9)
    logp = O.log_softmax(rng.standard_normal((3, 48, 41)).astype(np.float32) * 2)
    a = _search(logp, 10, 10)
    b = _search(logp, 10, 10, chunks=[(0, 1), (1, 2), (2, 17), (17, 48)])     # frame-by-frame then chunks
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_lm_decoder_surface_and_prologue():
    """lm_decoder.DecodeOptions/DecodeResource/BrainSpeechDecoder/DecodeNumpy (lm_decoder.cc:14-75): prologue vs oracle,
    decode vs oracle beam, ProcessBlank text rule, WFST paths refuse loudly."""
    import lm_decoder
    rng = np.random.default_rng(4)
    logits = rng.standard_normal((50, 41)).astype(np.float32) * 3
    pri = (rng.standard_normal((50, 41)) * 0.1).astype(np.float32)
    opts = lm_decoder.DecodeOptions(7000, 200, 17., 8., 0.325, 1.0, 0., 100)
    res = lm_decoder.DecodeResource("", "", "", "", "")
    dec = lm_decoder.BrainSpeechDecoder(res, opts)
    lm_decoder.DecodeNumpy(dec, logits, pri, math.log(90.0))
    dec.FinishDecoding()
    lp = O.lm_prologue(logits, pri, math.log(90.0))
    ref = O.prefix_beam_search(lp, 10, 10)
    out = dec.result()
    assert dec.DecodedSomething() == bool(ref[0][0])
    assert tuple(int(t) for t in out[0].tokens) == tuple(ref[0][0])
    assert abs(out[0].lm_score - ref[0][1]) < 1e-3 * max(1, abs(ref[0][1]))
    assert out[0].sentence == " ".join(str(t) for t in ref[0][0])
    dec.Reset()
    lm_decoder.DecodeNumpyLogProbs(dec, lp)
    assert tuple(int(t) for t in dec.result()[0].tokens) == tuple(ref[0][0])
    assert lm_decoder.process_blank("▁HELLO▁▁WORLD▁") == "hello world"
    with pytest.raises(FileNotFoundError):          # a graph path is read (tests/test_gpu_wfst.py decodes with one)
        lm_decoder.DecodeResource("/nonexistent/TLG.fst", "", "", "", "")
    with pytest.raises(RuntimeError, match="WFST"):
        dec.Rescore()                                # lattice rescoring belongs to the graph searcher


# ---- n-gram fusion (b2t_prefix_beam_search_lm_f32) ---------------------------------------------------------------
WORDS41 = [None] + [f"p{i}" for i in range(1, 41)]


def _search_lm(logp, lm, alpha, beta, first, second, chunks=None, eos=False):
    import ctypes as C
    import b2t_native as N
    import b2t_ops as ops
    dev = torch.device("cuda:0")
    lib = N.load()
    U, T, Cc = logp.shape
    L, NN = T + 1, T * second + 2
    state = torch.empty((lib.b2t_beam_state_bytes(L, NN) * U,), dtype=torch.uint8, device=dev)
    N.check(lib.b2t_beam_reset(ops._p(state), U, L, NN, ops._stream()), "reset")
    hyps = torch.zeros((U, second, L), dtype=torch.int32, device=dev)
    hl = torch.empty((U, second), dtype=torch.int32, device=dev)
    sc = torch.empty((U, second), device=dev); vs = torch.empty((U, second), device=dev); lms = torch.empty((U, second), device=dev)
    tm = torch.zeros((U, second, L), dtype=torch.int32, device=dev)
    d = lm.to_device(dev)
    for a, b in (chunks or [(0, T)]):
        lp = torch.from_numpy(np.ascontiguousarray(logp[:, a:b])).to(dev)
        N.check(lib.b2t_prefix_beam_search_lm_f32(
            ops._p(lp), None, U, b - a, Cc, first, second, 0, ops._p(state), L, NN, ops._p(hyps), ops._p(hl), ops._p(sc),
            ops._p(vs), ops._p(tm), ops._p(d["child"]), ops._p(d["logp"]), ops._p(d["bow"]), ops._p(d["suffix"]),
            ops._p(d["nstate"]), lm.V, lm.start_state, lm.eos if eos else -1, float(alpha), float(beta), float(lm.unk_logp),
            ops._p(lms), ops._stream()), "search_lm")
    flag = C.c_int(0)
    N.check(lib.b2t_beam_overflowed(ops._p(state), U, L, NN, C.byref(flag), ops._stream()), "ovf")
    assert flag.value == 0
    return hyps.cpu().numpy(), hl.cpu().numpy(), sc.cpu().numpy(), lms.cpu().numpy()


@pytest.mark.parametrize("order,first,second", [(2, 10, 10), (3, 10, 10), (5, 8, 16), (3, 10, 100)])
def test_prefix_beam_lm_vs_oracle(order, first, second):
    import ngram_lm
    text = ngram_lm.synthetic_arpa(WORDS41, order, 400, seed=order)
    lm = ngram_lm.NGramLM.from_arpa(text, WORDS41)
    o, tab = O.parse_arpa(text)
    rng = np.random.default_rng(order)
    U, T, C = 5, 50, 41
    logits = rng.standard_normal((U, T, C)).astype(np.float32) * 2.5
    logits[..., 0] += 1.0
    logp = O.log_softmax(logits)
    alpha, beta = 0.7, 0.3
    hyps, hl, sc, lms = _search_lm(logp, lm, alpha, beta, first, second)
    for u in range(U):
        ref = O.prefix_beam_search_lm(logp[u], o, tab, WORDS41, alpha, beta, first, second)
        got = [(tuple(hyps[u, i, :hl[u, i]]), sc[u, i], lms[u, i]) for i in range(second) if hl[u, i] >= 0]
        # compare as sets over the hypotheses whose total score is clear of the pruning boundary (fp32 vs fp64 ties)
        assert got[0][0] == ref[0][0]
        refd = {p: (a, l) for p, a, l, _ in ref}
        hits = 0
        for p, a, l in got:
            if p in refd:
                hits += 1
                np.testing.assert_allclose(a, refd[p][0], rtol=2e-4, atol=2e-4)
                np.testing.assert_allclose(l, refd[p][1], rtol=2e-4, atol=2e-4)
        assert hits >= len(ref) - max(2, len(ref) // 20)


def test_prefix_beam_lm_streaming_equals_offline_and_eos():
    import ngram_lm
    text = ngram_lm.synthetic_arpa(WORDS41, 3, 300, seed=9)
    lm = ngram_lm.NGramLM.from_arpa(text, WORDS41)
    rng = np.random.default_rng(4)
    logp = O.log_softmax((rng.standard_normal((3, 48, 41)) * 2.0).astype(np.float32))
    a = _search_lm(logp, lm, 0.5, 0.1, 10, 10)
    b = _search_lm(logp, lm, 0.5, 0.1, 10, 10, chunks=[(0, 7), (7, 8), (8, 30), (30, 48)])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    c = _search_lm(logp, lm, 0.5, 0.1, 10, 10, eos=True)
    for u in range(3):
        for i in range(10):
            if c[1][u, i] < 0:
                continue
            ids = c[0][u, i, :c[1][u, i]]
            np.testing.assert_allclose(c[3][u, i], 0.5 * lm.sentence_logp(ids, bos=True, eos=True) + 0.1 * len(ids), rtol=1e-4, atol=1e-4)


def test_streaming_full_size_equals_offline():
    """BASELINE configs[4] at size: 32 concurrent utterances, 5-gram LM, fed one frame per call (T = 200 calls) == one
    offline call, exactly (hypotheses, lengths, CTC and LM scores); the best hypothesis of two utterances matches the oracle."""
    import ngram_lm
    text = ngram_lm.synthetic_arpa(WORDS41, 5, 2000, seed=5)
    lm = ngram_lm.NGramLM.from_arpa(text, WORDS41)
    o, tab = O.parse_arpa(text)
    rng = np.random.default_rng(44)
    U, T = 32, 200
    logits = (rng.standard_normal((U, T, 41)) * 2.5).astype(np.float32)
    logits[..., 0] += 1.5
    logp = O.log_softmax(logits)
    a = _search_lm(logp, lm, 0.6, 0.2, 10, 10)
    b = _search_lm(logp, lm, 0.6, 0.2, 10, 10, chunks=[(t, t + 1) for t in range(T)])
    for x, y in zip(a[1:], b[1:]):                       # lengths, CTC scores, LM scores
        np.testing.assert_array_equal(x, y)
    valid = np.arange(a[0].shape[2])[None, None, :] < a[1][:, :, None]   # entries beyond a hypothesis' length are scratch
    np.testing.assert_array_equal(np.where(valid, a[0], 0), np.where(valid, b[0], 0))
    for u in (0, 17):
        ref = O.prefix_beam_search_lm(logp[u], o, tab, WORDS41, 0.6, 0.2, 10, 10)
        assert tuple(a[0][u, 0, :a[1][u, 0]]) == ref[0][0]
        np.testing.assert_allclose(a[2][u, 0], ref[0][1], rtol=2e-4, atol=2e-4)


def test_lm_decoder_with_token_lm():
    import lm_decoder
    import ngram_lm
    text = ngram_lm.synthetic_arpa(WORDS41, 3, 300, seed=2)
    lm = ngram_lm.NGramLM.from_arpa(text, WORDS41)
    o, tab = O.parse_arpa(text)
    res = lm_decoder.DecodeResource("", "", "", "", "")
    res.set_token_lm(lm)
    opts = lm_decoder.DecodeOptions(7000, 200, 17.0, 8.0, 0.5, 1.0, 0.0, 10)
    opts.lm_alpha, opts.lm_beta = 0.8, 0.2
    dec = lm_decoder.BrainSpeechDecoder(res, opts)
    rng = np.random.default_rng(8)
    logp = O.log_softmax((rng.standard_normal((40, 41)) * 2.0).astype(np.float32))
    lm_decoder.DecodeNumpyLogProbs(dec, logp)
    dec.FinishDecoding()
    ref = O.prefix_beam_search_lm(logp, o, tab, WORDS41, 0.8, 0.2, 10, 10)
    best = dec.result()[0]
    assert tuple(best.tokens) == ref[0][0]
    assert best.lm_score == pytest.approx(ref[0][2], rel=2e-4, abs=2e-4)
    assert best.ac_score == pytest.approx(ref[0][1] / dec.acoustic_scale, rel=2e-4, abs=2e-4)


# ---- word level: pronunciation lexicon + word n-gram (b2t_prefix_beam_search_lex_f32) ------------------------------
def _search_lex(logp, lex, lm, alpha, beta, first, second, chunks=None, eos=True):
    import ctypes as C
    import b2t_native as N
    import b2t_ops as ops
    dev = torch.device("cuda:0")
    lib = N.load()
    U, T, Cc = logp.shape
    L, NN = T + 1, T * second + 2
    state = torch.empty((lib.b2t_beam_state_bytes(L, NN) * U,), dtype=torch.uint8, device=dev)
    N.check(lib.b2t_beam_reset(ops._p(state), U, L, NN, ops._stream()), "reset")
    hyps = torch.zeros((U, second, L), dtype=torch.int32, device=dev)
    hl = torch.empty((U, second), dtype=torch.int32, device=dev)
    sc = torch.empty((U, second), device=dev); vs = torch.empty((U, second), device=dev); lms = torch.empty((U, second), device=dev)
    tm = torch.zeros((U, second, L), dtype=torch.int32, device=dev)
    dl, dm = lex.to_device(dev), lm.to_device(dev)
    d = N.LexLmDesc(dl["child"].data_ptr(), dl["wbeg"].data_ptr(), dl["wend"].data_ptr(), dl["wlist"].data_ptr(),
                    dm["cb"].data_ptr(), dm["ce"].data_ptr(), dm["ctok"].data_ptr(), dm["cnode"].data_ptr(),
                    dm["logp"].data_ptr(), dm["bow"].data_ptr(), dm["suffix"].data_ptr(), dm["nstate"].data_ptr(),
                    lm.start_state, lm.eos if eos else -1, 1, float(alpha), float(beta), float(lm.unk_logp))
    for a, b in (chunks or [(0, T)]):
        lp = torch.from_numpy(np.ascontiguousarray(logp[:, a:b])).to(dev)
        N.check(lib.b2t_prefix_beam_search_lex_f32(ops._p(lp), None, U, b - a, Cc, first, second, 0, ops._p(state), L, NN,
                                                   ops._p(hyps), ops._p(hl), ops._p(sc), ops._p(vs), ops._p(tm), C.byref(d),
                                                   ops._p(lms), ops._stream()), "search_lex")
    flag = C.c_int(0)
    N.check(lib.b2t_beam_overflowed(ops._p(state), U, L, NN, C.byref(flag), ops._stream()), "ovf")
    assert flag.value == 0
    return hyps.cpu().numpy(), hl.cpu().numpy(), sc.cpu().numpy(), lms.cpu().numpy()


@pytest.mark.parametrize("nwords,order", [(60, 2), (400, 3)])
def test_prefix_beam_lexicon_vs_oracle(nwords, order):
    import ngram_lm
    Cc = 14
    prons = ngram_lm.synthetic_lexicon(nwords, Cc, seed=nwords)
    lex = ngram_lm.Lexicon(prons, Cc)
    text = ngram_lm.synthetic_word_arpa(lex.words, order, 3 * nwords, seed=order)
    o, tab = O.parse_arpa(text)
    lm = ngram_lm.SparseNGramLM.from_arpa(text, lex.words)
    rng = np.random.default_rng(nwords + order)
    U, T = 4, 45
    logits = (rng.standard_normal((U, T, Cc)) * 1.5).astype(np.float32)
    logits[..., 1] += 0.7                                    # SIL a bit more likely: more complete words
    logp = O.log_softmax(logits)
    alpha, beta, first, second = 0.6, 0.5, 8, 16
    hyps, hl, sc, lms = _search_lex(logp, lex, lm, alpha, beta, first, second)
    for u in range(U):
        ref = O.prefix_beam_search_lexicon(logp[u], prons, o, tab, alpha, beta, first, second)
        got = {}
        for i in range(second):
            if hl[u, i] >= 0 and np.isfinite(lms[u, i]):
                got[tuple(hyps[u, i, :hl[u, i]])] = (sc[u, i], lms[u, i])
        assert ref, "no complete hypothesis in the oracle"
        best = max(got.items(), key=lambda kv: kv[1][0] + kv[1][1])
        assert best[0] == ref[0][0]
        hits = 0
        for p, words, ctc, lmscore, total in ref:
            if p in got:
                hits += 1
                np.testing.assert_allclose(got[p][0], ctc, rtol=2e-4, atol=2e-4)
                np.testing.assert_allclose(got[p][1], lmscore, rtol=2e-4, atol=3e-4)
                w2, l2 = ngram_lm.replay_words(lex, lm, p, alpha, beta)
                assert tuple(w2) == words
        assert hits >= len(ref) - 2


def test_prefix_beam_lexicon_streaming_equals_offline():
    import ngram_lm
    Cc = 14
    prons = ngram_lm.synthetic_lexicon(200, Cc, seed=5)
    lex = ngram_lm.Lexicon(prons, Cc)
    text = ngram_lm.synthetic_word_arpa(lex.words, 3, 500, seed=6)
    lm = ngram_lm.SparseNGramLM.from_arpa(text, lex.words)
    rng = np.random.default_rng(7)
    logp = O.log_softmax((rng.standard_normal((3, 40, Cc)) * 1.5).astype(np.float32))
    a = _search_lex(logp, lex, lm, 0.5, 0.3, 8, 12)
    b = _search_lex(logp, lex, lm, 0.5, 0.3, 8, 12, chunks=[(0, 1), (1, 9), (9, 10), (10, 40)])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_lm_decoder_with_lexicon():
    import lm_decoder
    import ngram_lm
    Cc = 14
    prons = ngram_lm.synthetic_lexicon(150, Cc, seed=21)
    lex = ngram_lm.Lexicon(prons, Cc)
    text = ngram_lm.synthetic_word_arpa(lex.words, 3, 400, seed=22)
    o, tab = O.parse_arpa(text)
    lm = ngram_lm.SparseNGramLM.from_arpa(text, lex.words)
    res = lm_decoder.DecodeResource("", "", "", "", "")
    res.set_lexicon_lm(lex, lm, sil=1)
    opts = lm_decoder.DecodeOptions(7000, 200, 17.0, 8.0, 0.5, 1.0, 0.0, 10)
    opts.lm_alpha, opts.lm_beta, opts.lm_eos, opts.first_beam_size, opts.second_beam_size = 0.6, 0.5, True, 8, 16
    dec = lm_decoder.BrainSpeechDecoder(res, opts)
    rng = np.random.default_rng(23)
    logits = (rng.standard_normal((45, Cc)) * 1.5).astype(np.float32); logits[:, 1] += 0.7
    logp = O.log_softmax(logits)
    lm_decoder.DecodeNumpyLogProbs(dec, logp)
    ref = O.prefix_beam_search_lexicon(logp, prons, o, tab, 0.6, 0.5, 8, 16)
    best = dec.result()[0]
    assert tuple(best.tokens) == ref[0][0]
    assert best.sentence == " ".join(ref[0][1]).lower()
    assert best.total_score == pytest.approx(ref[0][4], rel=2e-4, abs=3e-4)
