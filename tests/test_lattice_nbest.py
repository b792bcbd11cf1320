"""Host n-best of the pruned token lattice (csrc/lattice.cpp: b2t_lattice_nbest_host -- FinalizeSearch's
DeterminizeLatticePruned + ShortestPath(nbest), ctc_wfst_beam_search.cc:123-160) against the oracle's restatement
(oracle/wfst_oracle.py: nbest_word_sequences) -- no GPU involved.  Random acyclic lattices with epsilon-output arcs, and a
real lattice the GPU search left behind for one utterance of the tools/bench_wfst.py workload (tests/golden/wfst_lattice_u2.npz,
written by tools/experimental/dump_lattice.py: 4868 states, 9329 arcs, 217 final states)."""
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
