#!/usr/bin/env python3
"""Decode-side latency on one MI355X (BASELINE.json configs 4-5, LM-free and with a synthetic token-level n-gram LM;
the reference's word-level LMs/graphs are not in the checkout):
  offline   : U utterances x T frames in one call           -> ms per utterance
  streaming : 32 concurrent utterances fed one logit row (= one 80 ms patch frame) per call -> p50 / p95 ms per frame
Includes the DecodeNumpy prologue (log-softmax - priors - blank penalty) in the streaming loop."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_native as N, b2t_ops as ops, ngram_lm
lib = N.load(); dev = torch.device("cuda:0"); _p = ops._p
Cc, T, U = 41, 120, 32
WORDS = [None] + [f"p{i}" for i in range(1, 41)]
rng = np.random.default_rng(0)
logits = torch.from_numpy((rng.standard_normal((U, T, Cc)) * 3.0).astype(np.float32)).to(dev)
priors = torch.zeros((U, T, Cc), device=dev)

def run(order, first, second, n_per_order=20000):
    lm = None
    if order:
        lm = ngram_lm.NGramLM.from_arpa(ngram_lm.synthetic_arpa(WORDS, order, n_per_order, seed=order), WORDS)
        d = lm.to_device(dev)
    L, NN = T + 1, T * second + 2
    state = torch.empty((lib.b2t_beam_state_bytes(L, NN) * U,), dtype=torch.uint8, device=dev)
    hyps = torch.zeros((U, second, L), dtype=torch.int32, device=dev); hl = torch.empty((U, second), dtype=torch.int32, device=dev)
    sc = torch.empty((U, second), device=dev); vs = torch.empty((U, second), device=dev); lms = torch.empty((U, second), device=dev)
    tm = torch.zeros((U, second, L), dtype=torch.int32, device=dev)
    lp = torch.empty_like(logits)
    def search(x, nt):
        if lm is None:
            N.check(lib.b2t_prefix_beam_search_f32(_p(x), None, U, nt, Cc, first, second, 0, _p(state), L, NN, _p(hyps), _p(hl),
                                                   _p(sc), _p(vs), _p(tm), ops._stream()), "s")
        else:
            N.check(lib.b2t_prefix_beam_search_lm_f32(_p(x), None, U, nt, Cc, first, second, 0, _p(state), L, NN, _p(hyps), _p(hl),
                                                      _p(sc), _p(vs), _p(tm), _p(d["child"]), _p(d["logp"]), _p(d["bow"]),
                                                      _p(d["suffix"]), _p(d["nstate"]), lm.V, lm.start_state, -1, 0.6, 0.2,
                                                      float(lm.unk_logp), _p(lms), ops._stream()), "s")
    # offline
    ts = []
    for rep in range(5):
        N.check(lib.b2t_beam_reset(_p(state), U, L, NN, ops._stream()), "r")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N.check(lib.b2t_lm_prologue_f32(_p(logits), _p(priors), float(np.log(90.0)), _p(lp), U * T, Cc, ops._stream()), "p")
        search(lp, T)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    off = min(ts) * 1e3
    # streaming: one frame per call for all 32 utterances
    N.check(lib.b2t_beam_reset(_p(state), U, L, NN, ops._stream()), "r")
    frame = torch.empty((U, 1, Cc), device=dev); pri1 = torch.zeros((U, 1, Cc), device=dev); lp1 = torch.empty((U, 1, Cc), device=dev)
    lat = []
    for t in range(T):
        frame.copy_(logits[:, t:t + 1])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N.check(lib.b2t_lm_prologue_f32(_p(frame), _p(pri1), float(np.log(90.0)), _p(lp1), U, Cc, ops._stream()), "p")
        search(lp1, 1)
        best = hl[:, 0].cpu()      # the host reads the running best hypothesis length every frame
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat[5:]) * 1e3
    name = f"{order}-gram ({lm.n_nodes} nodes)" if lm else "no LM"
    print(f"{name:24s} beams {first}/{second}: offline {off:7.3f} ms for {U} x {T} frames ({off / U:6.3f} ms/utterance, "
          f"{off / T * 1e3:6.1f} us/frame) | streaming {U} utterances: p50 {np.percentile(lat, 50):6.3f} ms/frame, p95 {np.percentile(lat, 95):6.3f}")

for order, first, second in ((0, 10, 10), (3, 10, 10), (5, 10, 10), (5, 16, 32), (3, 10, 100)):   # last: BASELINE configs[3]
    run(order, first, second)


def run_words(n_words, order, first, second, n_per_order):
    """Word level: synthetic pronunciation lexicon (n_words) + synthetic word n-gram; logits that spell random word
    sequences (so that complete words exist), otherwise as above."""
    t0 = time.time()
    prons = ngram_lm.synthetic_lexicon(n_words, Cc, seed=n_words)
    lex = ngram_lm.Lexicon(prons, Cc)
    lm = ngram_lm.SparseNGramLM.from_arpa(ngram_lm.synthetic_word_arpa(lex.words, order, n_per_order, seed=order), lex.words)
    build_s = time.time() - t0
    rs = np.random.RandomState(1)
    lg = np.full((U, T, Cc), -2.0, dtype=np.float32)
    for u in range(U):
        t = 0
        while t < T - 12:
            for c in list(prons[lex.words[rs.randint(n_words)]][0]) + [1]:
                lg[u, t, c] = 4.0; lg[u, t + 1, 0] = 3.0; t += 2
                if t >= T - 2: break
        lg[u] += rs.standard_normal((T, Cc)).astype(np.float32) * 0.7
    lgt = torch.from_numpy(lg).to(dev)
    L, NN = T + 1, T * second + 2
    state = torch.empty((lib.b2t_beam_state_bytes(L, NN) * U,), dtype=torch.uint8, device=dev)
    hyps = torch.zeros((U, second, L), dtype=torch.int32, device=dev); hl = torch.empty((U, second), dtype=torch.int32, device=dev)
    sc = torch.empty((U, second), device=dev); vs = torch.empty((U, second), device=dev); lms = torch.empty((U, second), device=dev)
    tm = torch.zeros((U, second, L), dtype=torch.int32, device=dev)
    dl, dm = lex.to_device(dev), lm.to_device(dev)
    d = N.LexLmDesc(dl["child"].data_ptr(), dl["wbeg"].data_ptr(), dl["wend"].data_ptr(), dl["wlist"].data_ptr(),
                    dm["cb"].data_ptr(), dm["ce"].data_ptr(), dm["ctok"].data_ptr(), dm["cnode"].data_ptr(),
                    dm["logp"].data_ptr(), dm["bow"].data_ptr(), dm["suffix"].data_ptr(), dm["nstate"].data_ptr(),
                    lm.start_state, lm.eos, 1, 0.6, 0.5, float(lm.unk_logp))
    lp = torch.empty_like(lgt); pri = torch.zeros_like(lgt)
    def search(x, nt):
        N.check(lib.b2t_prefix_beam_search_lex_f32(_p(x), None, U, nt, Cc, first, second, 0, _p(state), L, NN, _p(hyps), _p(hl),
                                                   _p(sc), _p(vs), _p(tm), C.byref(d), _p(lms), ops._stream()), "s")
    ts = []
    for rep in range(5):
        N.check(lib.b2t_beam_reset(_p(state), U, L, NN, ops._stream()), "r")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N.check(lib.b2t_lm_prologue_f32(_p(lgt), _p(pri), 0.0, _p(lp), U * T, Cc, ops._stream()), "p")
        search(lp, T)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    off = min(ts) * 1e3
    done = int((torch.isfinite(lms[:, 0])).sum())
    N.check(lib.b2t_beam_reset(_p(state), U, L, NN, ops._stream()), "r")
    frame = torch.empty((U, 1, Cc), device=dev); pri1 = torch.zeros((U, 1, Cc), device=dev); lp1 = torch.empty((U, 1, Cc), device=dev)
    lat = []
    for t in range(T):
        frame.copy_(lgt[:, t:t + 1])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N.check(lib.b2t_lm_prologue_f32(_p(frame), _p(pri1), 0.0, _p(lp1), U, Cc, ops._stream()), "p")
        search(lp1, 1)
        best = hl[:, 0].cpu()
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat[5:]) * 1e3
    mb = (lex.child.nbytes + lm.ctok.nbytes * 2 + lm.logp.nbytes * 4 + lm.cb.nbytes * 2) / 1e6
    print(f"words: {n_words} ({lex.n_nodes} trie nodes), {order}-gram ({lm.n_nodes} nodes), tables {mb:.0f} MB, host build {build_s:.1f} s, "
          f"beams {first}/{second}: offline {off:7.3f} ms for {U} x {T} frames ({off / U:6.3f} ms/utterance) | streaming: p50 "
          f"{np.percentile(lat, 50):6.3f} ms/frame, p95 {np.percentile(lat, 95):6.3f} | best hypothesis complete for {done}/{U}")


run_words(20000, 3, 10, 16, 200000)
run_words(125000, 3, 10, 16, 500000)
