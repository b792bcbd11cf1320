"""Host graph compiler (csrc/graphc.cpp through include/b2t.h b2t_fst_*): the FST algebra of the reference's
recipe, language_model/tools/fst/make_tlg.sh:29-46, checked against DEFINITIONS (OpenFST / Kaldi binaries are not in the
image, so no output file of theirs exists to compare with):
  * composition and trim equal the Python restatements (wfst.compose / wfst.trim) state for state, arc for arc;
  * determinize-star output is deterministic* and carries the same weighted relation {(input string, output string) -> weight}
    as its input (weights of equal pairs combined by min in the tropical and by log-add in the log semiring);
  * minimize-encoded keeps the relation and reaches the Myhill-Nerode partition computed by brute force;
  * the optimised T o det-min(L o G) assigns every phone sequence the same best (word sequence, cost) as the plain graph;
  * OpenFST container: written by C++ / read by Python and the other way round; grammar scores equal the Python walk.
CPU only."""
import math
import os

import numpy as np
import pytest

import ngram_lm
import wfst

EPS = 0


def rand_fst(rs, n, n_arcs, n_il, n_ol, eps_in=0.2, eps_out=0.3, n_final=2):
    f = wfst.Fst()
    for _ in range(n):
        f.add_state()
    f.start = 0
    for _ in range(n_arcs):
        s, d = int(rs.randint(n)), int(rs.randint(n))
        il = EPS if rs.rand() < eps_in else int(rs.randint(1, n_il + 1))
        ol = EPS if rs.rand() < eps_out else int(rs.randint(1, n_ol + 1))
        f.add_arc(s, il, ol, float(np.float32(rs.rand() * 2)), d)
    for s in rs.choice(n, size=n_final, replace=False):
        f.final[int(s)] = float(np.float32(rs.rand()))
    return f


def same_fst(h: wfst.HostFst, f: wfst.Fst):
    g = h.to_fst()
    assert g.n == f.n and g.start == f.start
    by_state = lambda arcs: sorted(arcs, key=lambda a: a[0])          # stable: arc order within a state is compared too
    ga, fa = by_state(g.arcs), by_state(f.arcs)
    assert len(ga) == len(fa)
    for x, y in zip(ga, fa):
        assert x[:3] == y[:3] and x[4] == y[4] and abs(x[3] - y[3]) < 1e-5, (x, y)
    assert set(g.final) == set(f.final) and all(abs(g.final[s] - f.final[s]) < 1e-5 for s in f.final)


@pytest.mark.parametrize("seed", range(6))
def test_compose_and_trim_equal_the_python_restatement(seed):
    rs = np.random.RandomState(seed)
    a = rand_fst(rs, 7, 22, 4, 5)
    b = rand_fst(rs, 6, 20, 5, 4)
    c_py = wfst.compose(a, b)
    c_nat = wfst.HostFst.from_fst(a).compose(wfst.HostFst.from_fst(b))
    same_fst(c_nat, c_py)
    try:
        t_py = wfst.trim(c_py)
    except ValueError:
        with pytest.raises(RuntimeError):
            c_nat.trim()
        return
    same_fst(c_nat.trim(), t_py)


def relation(f: wfst.Fst, max_in: int, log_sr: bool, max_steps: int = 14):
    """{(input string, output string): weight} over all accepting paths with <= max_in input symbols and <= max_steps arcs;
    weights of equal pairs combined by min / log-add."""
    out = f.out()
    rel = {}
    stack = [(f.start, (), (), 0.0, 0)]
    while stack:
        s, i, o, w, k = stack.pop()
        if s in f.final:
            key, tot = (i, o), w + f.final[s]
            if key not in rel:
                rel[key] = tot
            elif log_sr:
                m = min(rel[key], tot)
                rel[key] = m - math.log1p(math.exp(-abs(rel[key] - tot)))
            else:
                rel[key] = min(rel[key], tot)
        if k >= max_steps:
            continue
        for il, ol, aw, d in out[s]:
            i2 = i + (il,) if il else i
            if len(i2) > max_in:
                continue
            stack.append((d, i2, o + (ol,) if ol else o, w + aw, k + 1))
    return rel


def toy_LG(seed, n_words=9, homophones=True):
    rs = np.random.RandomState(seed)
    phones = [2, 3, 4, 5]
    prons = {}
    base = []
    for i in range(n_words):
        if homophones and base and rs.rand() < 0.3:
            p = base[rs.randint(len(base))]
            if rs.rand() < 0.5:
                p = p[:max(1, len(p) - 1)]                  # a prefix of another pronunciation
        else:
            p = tuple(int(phones[j]) for j in rs.randint(len(phones), size=rs.randint(1, 4)))
            base.append(p)
        prons[f"w{i}"] = [p]
    words = sorted(prons)
    table = ["<eps>"] + words + ["#0", "<s>", "</s>"]
    word_id = {w: i for i, w in enumerate(table) if 0 < i <= len(words)}
    wd0 = len(words) + 1
    td0 = 10
    L, n_dis = wfst.lexicon_fst_disambig(prons, word_id, 0.4, 1, td0, wd0)
    arpa = ngram_lm.synthetic_word_arpa(words, 2, 12, seed=seed + 50)
    G = wfst.grammar_fst(arpa, word_id, wd0)
    return L, G, n_dis


def check_determinized_star(f: wfst.Fst):
    out = f.out()
    for s in range(f.n):
        ils = [a[0] for a in out[s]]
        nz = [x for x in ils if x != 0]
        assert len(nz) == len(set(nz)), f"state {s}: two arcs with the same input label"
        if 0 in ils:          # epsilon-input arcs only from non-final states with exactly one arc (determinize-star.h:39-46)
            assert len(ils) == 1 and s not in f.final, f"state {s}: an epsilon-input arc next to other arcs / on a final state"


@pytest.mark.parametrize("seed,use_log", [(s, l) for s in range(5) for l in (False, True)])
def test_determinize_star_keeps_the_weighted_relation(seed, use_log):
    L, G, _ = toy_LG(seed)
    LG = wfst.HostFst.from_fst(L).compose(wfst.HostFst.from_fst(G).arcsort()).trim()
    det = LG.determinize_star(use_log=use_log)
    fd = det.to_fst()
    check_determinized_star(fd)
    r0 = relation(LG.to_fst(), 7, use_log, max_steps=18)
    r1 = relation(fd, 7, use_log, max_steps=30)
    assert len(r0) > 20 and set(r0) == set(r1)
    for k in r0:
        assert abs(r0[k] - r1[k]) < 1e-3, (k, r0[k], r1[k])


def test_determinize_star_refuses_non_functional_input():
    f = wfst.Fst()
    for _ in range(3):
        f.add_state()
    f.start = 0
    f.add_arc(0, 1, 5, 0.5, 1); f.add_arc(0, 1, 6, 0.25, 2)          # same input, two outputs
    f.final[1] = 0.0; f.final[2] = 0.0
    with pytest.raises(RuntimeError, match="not functional"):
        wfst.HostFst.from_fst(f).determinize_star()


def nerode_classes(f: wfst.Fst, delta=1.0 / 1024):
    """Brute force: states are distinguishable if their (quantised) final weights differ or some (il, ol, quantised weight)
    leads to distinguishable states / exists on one side only; iterate to the fixed point over all pairs."""
    q = lambda w: round(w / delta)
    out = f.out()
    tr = [{(il, ol, q(w)): d for il, ol, w, d in out[s]} for s in range(f.n)]
    fin = [q(f.final[s]) if s in f.final else None for s in range(f.n)]
    dist = [[fin[a] != fin[b] or set(tr[a]) != set(tr[b]) for b in range(f.n)] for a in range(f.n)]
    changed = True
    while changed:
        changed = False
        for a in range(f.n):
            for b in range(a):
                if not dist[a][b] and any(dist[tr[a][k]][tr[b][k]] for k in tr[a]):
                    dist[a][b] = dist[b][a] = True; changed = True
    reps = []
    for s in range(f.n):
        if not any(not dist[s][r] for r in reps):
            reps.append(s)
    return len(reps)


@pytest.mark.parametrize("seed", range(5))
def test_minimize_encoded_is_minimal_and_keeps_the_relation(seed):
    L, G, _ = toy_LG(seed, n_words=7)
    det = wfst.HostFst.from_fst(L).compose(wfst.HostFst.from_fst(G).arcsort()).trim().determinize_star(use_log=True).trim()
    mn = det.minimize_encoded()
    fd, fm = det.to_fst(), mn.to_fst()
    assert fm.n <= fd.n
    assert fm.n == nerode_classes(fd)
    check_determinized_star(fm)
    r0, r1 = relation(fd, 7, False, max_steps=30), relation(fm, 7, False, max_steps=30)
    assert len(r0) > 20
    assert set(r0) == set(r1)
    for k in r0:
        assert abs(r0[k] - r1[k]) < 0.02            # weights quantised to 1/1024 per arc
    assert mn.minimize_encoded().info()["n_states"] == fm.n      # idempotent


def best_path_cost(g: "wfst.DecodeGraph", phones):
    """Cheapest (cost, words) of the graph for a phone-token sequence read greedily (every frame one token, no acoustic
    scores): Viterbi over the CSR arrays with epsilon closure."""
    INF = float("inf")
    cur = {g.start: (0.0, ())}

    def close(d):
        st = list(d)
        while st:
            s = st.pop()
            c, wds = d[s]
            for a in range(g.row[s], g.row[s] + g.n_eps[s]):
                n2, c2 = int(g.next[a]), c + float(g.weight[a])
                w2 = wds + (int(g.olabel[a]),) if g.olabel[a] else wds
                if n2 not in d or c2 < d[n2][0] - 1e-9:
                    d[n2] = (c2, w2); st.append(n2)
        return d

    cur = close(cur)
    for p in phones:
        nxt = {}
        for s, (c, wds) in cur.items():
            a0, a1 = g.row[s] + g.n_eps[s], g.row[s + 1]
            lo = a0 + int(np.searchsorted(g.ilabel[a0:a1], p, "left")); hi = a0 + int(np.searchsorted(g.ilabel[a0:a1], p, "right"))
            for a in range(lo, hi):
                n2, c2 = int(g.next[a]), c + float(g.weight[a])
                w2 = wds + (int(g.olabel[a]),) if g.olabel[a] else wds
                if n2 not in nxt or c2 < nxt[n2][0] - 1e-9:
                    nxt[n2] = (c2, w2)
        if not nxt:
            return INF, ()
        cur = close(nxt)
    best = (INF, ())
    for s, (c, wds) in cur.items():
        if np.isfinite(g.final[s]) and c + float(g.final[s]) < best[0]:
            best = (c + float(g.final[s]), wds)
    return best


def test_optimised_tlg_decodes_like_the_plain_graph():
    prons = ngram_lm.synthetic_lexicon(40, 41, seed=5)
    words = sorted(prons)
    arpa = ngram_lm.synthetic_word_arpa(words, 3, 150, seed=6)
    st0, st1 = {}, {}
    g_plain = wfst.build_tlg_native(prons, arpa, optimize=False, stats=st0)
    g_opt = wfst.build_tlg_native(prons, arpa, optimize=True, stats=st1)
    g_py = wfst.build_tlg(prons, arpa)
    assert st1["LG"]["n_states"] < st0["LG"]["n_states"] and g_opt.n_arcs < g_plain.n_arcs
    rs = np.random.RandomState(0)
    checked = 0
    for _ in range(25):
        seq = [words[i] for i in rs.randint(len(words), size=rs.randint(1, 4))]
        toks = []
        for w in seq:
            prev = None
            for c in list(prons[w][0]) + [1]:          # the word's phones, then SIL; token = class + 1; blank between repeats
                if c == prev:
                    toks.append(1)
                toks += [c + 1] * int(rs.randint(1, 3))
                prev = c
        c0, w0 = best_path_cost(g_plain, toks)
        c1, w1 = best_path_cost(g_opt, toks)
        c2, w2 = best_path_cost(g_py, toks)
        assert np.isfinite(c0)
        assert abs(c0 - c1) < 0.05 and abs(c0 - c2) < 1e-3, (seq, c0, c1, c2)
        if len({c0}) and w0 == w2:                      # (homophones tie: compare words only where the Python graph agrees)
            assert [g_plain.words[x] for x in w0] == [g_opt.words[x] for x in w1] or abs(c0 - c1) < 0.05
        checked += 1
    assert checked == 25


def test_openfst_container_and_grammar_score(tmp_path):
    rs = np.random.RandomState(3)
    f = rand_fst(rs, 9, 30, 4, 4)
    h = wfst.HostFst.from_fst(f)
    h.write_openfst(str(tmp_path / "a.fst"))
    same_fst(wfst.HostFst.from_fst(wfst.read_openfst_vector(str(tmp_path / "a.fst"))), f)      # C++ wrote, Python read
    wfst.write_openfst_vector(f, str(tmp_path / "b.fst"))
    same_fst(wfst.HostFst.read_openfst(str(tmp_path / "b.fst")), f)                             # Python wrote, C++ read
    with open(str(tmp_path / "bad.fst"), "wb") as fh:
        fh.write(b"not an fst")
    with pytest.raises(RuntimeError):
        wfst.HostFst.read_openfst(str(tmp_path / "bad.fst"))
    words = [f"w{i}" for i in range(12)]
    word_id = {w: i + 1 for i, w in enumerate(words)}
    wd0 = len(words) + 1
    G = wfst.grammar_fst(ngram_lm.synthetic_word_arpa(words, 3, 60, seed=9), word_id, wd0)
    Gh = wfst.HostFst.from_fst(G).arcsort()
    for _ in range(40):
        ids = [int(x) for x in rs.randint(1, len(words) + 1, size=rs.randint(0, 6))]
        a, b = wfst.grammar_score(G, ids, wd0), Gh.grammar_score(ids, wd0)
        assert (math.isinf(a) and math.isinf(b)) or abs(a - b) < 1e-4, (ids, a, b)


def test_prepare_lm_projects_non_acceptors_and_finds_the_backoff_label():
    """fst::ReadAndPrepareLmFst (kaldi/fstext/kaldi-fst-io.cc:129-147) in C++ (b2t_fst_prepare_lm; round 5: the Python host used to copy
    the whole grammar into numpy for the acceptor test): a transducer grammar (#0 on the back-off arcs' INPUT side only, as
    eps2disambig.pl leaves it) is projected on its output labels and arc-sorted, back-off label 0; an acceptor stays as it is; an
    acceptor with #0 on BOTH sides and no label-0 arc reports #0 as its back-off label -- against the numpy statement of the same rule."""
    rng = np.random.RandomState(3)
    n, m, dis = 7, 40, 9
    src = rng.randint(0, n, m).astype(np.int32); dst = rng.randint(0, n, m).astype(np.int32)
    ol = rng.randint(1, 8, m).astype(np.int32); w = rng.rand(m).astype(np.float32)
    fc = np.where(rng.rand(n) < 0.4, rng.rand(n), np.inf).astype(np.float32)
    back = rng.rand(m) < 0.25
    cases = {"transducer": (np.where(back, dis, ol).astype(np.int32), np.where(back, 0, ol).astype(np.int32)),
             "acceptor_eps": (np.where(back, 0, ol).astype(np.int32),) * 2,
             "acceptor_hash0": (np.where(back, dis, ol).astype(np.int32),) * 2}
    for name, (il_c, ol_c) in cases.items():
        f = wfst.HostFst.from_arrays(n, 0, src, il_c, ol_c, w, dst, fc)
        got, backoff = f.prepare_lm(dis)
        row, il, ol2, w2, nx, fc2 = got.arrays()
        # the rule, in numpy
        want_il = ol_c if np.any(il_c != ol_c) else il_c
        want_back = dis if (not np.any(il_c != ol_c) and not np.any(il_c == 0) and np.any(il_c == dis)) else 0
        assert backoff == want_back, name
        assert np.array_equal(fc2, fc)
        for s in range(n):
            idx = np.flatnonzero(src == s)
            order = idx[np.argsort(want_il[idx], kind="stable")]
            a, b = int(row[s]), int(row[s + 1])
            assert b - a == len(idx)
            assert np.array_equal(il[a:b], want_il[order]) and np.array_equal(ol2[a:b], ol_c[order]), (name, s)
            assert np.array_equal(nx[a:b], dst[order]) and np.array_equal(w2[a:b], w[order])
        assert f.prepare_lm(None)[1] == 0
