#!/usr/bin/env python3
"""Which part of the training step invalidates a stream capture: forward only, forward + loss, + backward, the update alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import bench, b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(10)
model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)
xs = ops.augment_smooth(x, 2, 100, "same", cut=1, white_std=1.0, offset_std=0.2, seed=3)
for i in range(4): ts.step(xs, days, labels, nts - 1, lens)
torch.cuda.synchronize()
def fwd():
    with torch.no_grad():
        return model(xs, days)
def fwd_grad():
    return model(xs, days)
def fwd_bwd():
    model(xs, days).sum().backward()
cases = {"augment": lambda: ops.augment_smooth(x, 2, 100, "same", cut=1, white_std=1.0, offset_std=0.2, seed=3),
         "forward (no grad)": fwd, "forward (autograd)": fwd_grad, "forward + backward": fwd_bwd,
         "compute_grads": lambda: ts.compute_grads(xs, days, labels, nts - 1, lens), "apply_update": ts.apply_update}
S_CAP = torch.cuda.Stream()
S_CAP.wait_stream(torch.cuda.current_stream())
for name, fn in cases.items():
    try:
        fn(); torch.cuda.synchronize()
        # (the executor's worker queues are calibrated per caller stream, with host synchronisation: warm up ON the capture stream)
        with torch.cuda.stream(S_CAP):
            fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=S_CAP, capture_error_mode="thread_local"):
            fn()
        torch.cuda.synchronize()
        import time
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        with torch.cuda.stream(S_CAP):
            for _ in range(10): fn()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{name}: captured; replay {(t1 - t0) * 100:.3f} ms, issued {(t2 - t1) * 100:.3f} ms per call", flush=True)
    except Exception as e:      # noqa: BLE001
        print(f"{name}: FAILED {repr(e)[:160]}", flush=True)
        torch.cuda.synchronize()
