import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_ops as ops
from rnn_model import GRUDecoder
dev = torch.device("cuda:0")
U, F, H, L, C, PATCH, STRIDE = 32, 512, 768, 5, 41, 14, 4
model = GRUDecoder(F, H, 4, C, 0.0, 0.0, L, PATCH, STRIDE).to(dev).eval()
day = torch.zeros(U, dtype=torch.int32, device=dev)
x = torch.randn(U, PATCH, F, device=dev) * 0.5
for mode in (True, False, True):
    ops.STREAM["graph"] = mode
    states = None
    ts, parts = [], []
    with torch.no_grad():
        for f in range(120):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            lg, states = model(x, day, states, True)
            t1 = time.perf_counter()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            ts.append(t2 - t0); parts.append(t1 - t0)
    print(f"graph={mode}: call+sync p50 {np.percentile(ts[10:], 50) * 1e3:.4f} ms, host part p50 {np.percentile(parts[10:], 50) * 1e3:.4f} ms;",
          {k[1]: (e['calls'], e['graph'] is not None) for k, e in model._graphs.items()})
