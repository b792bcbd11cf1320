#!/usr/bin/env python3
"""BASELINE configs[4] end to end: 32 concurrent utterances, one 80 ms patch frame per call (14 bins of 20 ms, stride 4):
GRU forward with carried state (shipped shape: H=768, 5 layers, patch 14/4) -> LM prologue -> 5-gram prefix beam, host
reads the running best hypothesis.  Prints p50 / p95 per frame for the GRU step alone and for the whole chain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_native as N, b2t_ops as ops, ngram_lm
from rnn_model import GRUDecoder
lib = N.load(); dev = torch.device("cuda:0"); _p = ops._p
U, F, H, L, C = 32, 512, int(os.environ.get("B2T_H", 768)), 5, 41
PATCH, STRIDE, NFR = 14, 4, 150
torch.manual_seed(0)
model = GRUDecoder(F, H, 4, C, 0.0, 0.0, L, PATCH, STRIDE).to(dev).eval()
day = torch.zeros(U, dtype=torch.int32, device=dev)
x_all = torch.randn(U, PATCH + STRIDE * (NFR - 1), F, device=dev) * 0.5
WORDS = [None] + [f"p{i}" for i in range(1, 41)]
lm = ngram_lm.NGramLM.from_arpa(ngram_lm.synthetic_arpa(WORDS, 5, 20000, seed=5), WORDS)
d = lm.to_device(dev)
first, second = 10, 10
Lm, NN = NFR + 1, NFR * second + 2
state = torch.empty((lib.b2t_beam_state_bytes(Lm, NN) * U,), dtype=torch.uint8, device=dev)
hyps = torch.zeros((U, second, Lm), dtype=torch.int32, device=dev); hl = torch.empty((U, second), dtype=torch.int32, device=dev)
sc = torch.empty((U, second), device=dev); vs = torch.empty((U, second), device=dev); lms = torch.empty((U, second), device=dev)
tm = torch.zeros((U, second, Lm), dtype=torch.int32, device=dev)
pri = torch.zeros((U, 1, C), device=dev); lp = torch.empty((U, 1, C), device=dev)
N.check(lib.b2t_beam_reset(_p(state), U, Lm, NN, ops._stream()), "r")
states = None
t_gru, t_all = [], []
# offline logits for the equivalence check (whole sequence in one pass)
with torch.no_grad():
    ref_logits, _ = model(x_all, day, None, True)
got = []
with torch.no_grad():
    for f in range(NFR):
        xf = x_all[:, f * STRIDE: f * STRIDE + PATCH].contiguous()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        logits, states = model(xf, day, states, True)          # [U, 1, C]
        torch.cuda.synchronize(); t1 = time.perf_counter()
        N.check(lib.b2t_lm_prologue_f32(_p(logits), _p(pri), float(np.log(90.0)), _p(lp), U, C, ops._stream()), "p")
        N.check(lib.b2t_prefix_beam_search_lm_f32(_p(lp), None, U, 1, C, first, second, 0, _p(state), Lm, NN, _p(hyps), _p(hl),
                                                  _p(sc), _p(vs), _p(tm), _p(d["child"]), _p(d["logp"]), _p(d["bow"]),
                                                  _p(d["suffix"]), _p(d["nstate"]), lm.V, lm.start_state, -1, 0.6, 0.2,
                                                  float(lm.unk_logp), _p(lms), ops._stream()), "s")
        best = hl[:, 0].cpu()
        t2 = time.perf_counter()
        t_gru.append(t1 - t0); t_all.append(t2 - t0); got.append(logits)
got = torch.cat(got, 1)
err = float((got - ref_logits).abs().max())
g, a = np.array(t_gru[10:]) * 1e3, np.array(t_all[10:]) * 1e3
print(f"H={H}, 5 layers, patch 14/4, {U} utterances, one 80 ms frame per call: GRU step p50 {np.percentile(g, 50):.3f} ms "
      f"(p95 {np.percentile(g, 95):.3f}); GRU + prologue + 5-gram beam + host read p50 {np.percentile(a, 50):.3f} ms "
      f"(p95 {np.percentile(a, 95):.3f}) = {np.percentile(a, 50) / 80 * 100:.2f} % of real time; "
      f"streamed logits vs one offline pass: max |diff| {err:.2e}")
