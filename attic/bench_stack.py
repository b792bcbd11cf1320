#!/usr/bin/env python3
"""Forward GRU stack at the bench shape: per-layer plan (mode 1, 6 chunks) vs one persistent launch (mode 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_ops as ops
from rnn_model import GRUDecoder
dev = torch.device("cuda:0")
B, T, F, H, L = int(os.environ.get("B2T_B", 64)), 500, 512, 512, 5
drop = float(os.environ.get("B2T_DROP", "0.4"))
torch.manual_seed(0)
model = GRUDecoder(F, H, 4, 41, drop, 0.0, L, 0, 0).to(dev).train()
x = torch.randn(B, T, F, device=dev) * 0.5
day = torch.zeros(B, dtype=torch.int32, device=dev)
prm, dims = model._kernel_params(), model._dims
for mode in (1, 4):
    ops.GRU_MODE["value"] = mode
    for _ in range(3):
        ops.model_forward(dims, prm, x, day, None, model._ws, True, 0.0, drop, seed=1, reuse_saved=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for i in range(n):
        ops.model_forward(dims, prm, x, day, None, model._ws, True, 0.0, drop, seed=2 + i, reuse_saved=True)
    e1.record(); torch.cuda.synchronize()
    model._ws.check_sync()
    print(f"mode {mode}: forward (day layer + stack + head, saved tensors) {e0.elapsed_time(e1) / n:.3f} ms")
if os.environ.get("B2T_LIB"):   # -DB2T_TIMING build: s_memtime ticks (100 MHz) per item, layer L/2, slice 0
    w = model._ws.sync_ws(0, T, dev, B, H, "stk")[8:24].cpu().tolist()
    names = ["drain", "poll", "stores", "mfma", "pubdrain", "B1|partials", "gates", "B2"]
    print("recurrent waves :", " ".join(f"{n}={w[i]}" for i, n in enumerate(names)), "sum", sum(w[:8]))
    print("projection waves:", " ".join(f"{n}={w[8 + i]}" for i, n in enumerate(names)), "sum", sum(w[8:16]))
