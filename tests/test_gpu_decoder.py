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
    with pytest.raises(NotImplementedError):
        lm_decoder.DecodeResource("TLG.fst", "", "", "words.txt", "")
    with pytest.raises(NotImplementedError):
        dec.Rescore()
