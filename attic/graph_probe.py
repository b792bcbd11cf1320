#!/usr/bin/env python3
"""Can a streaming call of the executor (serial plan, one stream) be captured into a hipGraph through torch.cuda.CUDAGraph, and what
does replaying it cost per frame?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_native as N, b2t_ops as ops
from rnn_model import GRUDecoder
dev = torch.device("cuda:0")
U, F, H, L, C, PATCH, STRIDE = 32, 512, 768, 5, 41, 14, 4
torch.manual_seed(0)
model = GRUDecoder(F, H, 4, C, 0.0, 0.0, L, PATCH, STRIDE).to(dev).eval()
day = torch.zeros(U, dtype=torch.int32, device=dev)
x_all = torch.randn(U, PATCH + STRIDE * 59, F, device=dev) * 0.5
sx = torch.zeros(U, PATCH, F, device=dev); ss = torch.zeros(L, U, H, device=dev)
with torch.no_grad():
    ref = []
    states = None
    for f in range(6):
        lg, states = model(x_all[:, f * STRIDE: f * STRIDE + PATCH].contiguous(), day, states, True)
        ref.append(lg.clone())
    ref_states = states.clone()
    # warm-up on a side stream, then capture one frame with carried state
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model(sx, day, ss, True)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_l, out_s = model(sx, day, ss, True)
    print("captured", flush=True)
    # replay: frame 0 needs h0 -> states None path is another graph; here start from the executor's first-frame state
    lg0, st = model(x_all[:, :PATCH].contiguous(), day, None, True)
    got = [lg0]
    for f in range(1, 6):
        sx.copy_(x_all[:, f * STRIDE: f * STRIDE + PATCH]); ss.copy_(st)
        g.replay()
        got.append(out_l.clone()); st = out_s.clone()
    torch.cuda.synchronize()
    print("max |dlogits|", max(float((a - b).abs().max()) for a, b in zip(got, ref)), "max |dstate|", float((st - ref_states).abs().max()))
    for name, fn in (("executor", lambda: model(sx, day, ss, True)), ("graph replay", lambda: g.replay())):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts = np.array(ts) * 1e3
        print(f"{name}: call+sync p50 {np.percentile(ts, 50):.4f} ms p95 {np.percentile(ts, 95):.4f}; back-to-back {e0.elapsed_time(e1) / 200:.4f} ms")
