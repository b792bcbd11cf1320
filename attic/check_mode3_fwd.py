#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_ops as ops
from rnn_model import GRUDecoder
dev = torch.device("cuda:0")
F, H, D, C, L, T, B = 512, 512, 4, 41, 2, 60, 64
torch.manual_seed(3)
x = torch.randn(B, T, F, device=dev); day = torch.randint(0, D, (B,), device=dev, dtype=torch.int32)
torch.manual_seed(5)
model = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0).to(dev).train()
res = {}
for mode in (1, 3, 1, 3):
    ops.GRU_MODE["value"] = mode
    logits, hidden, ctx = ops.model_forward(model._dims, model._kernel_params(), x, day, None, model._ws, save=True, reuse_saved=False)
    torch.cuda.synchronize()
    cur = dict(logits=logits.clone(), **{f"out{l}": ctx.outs[l].clone() for l in range(L)}, **{f"res{l}": ctx.reserves[l].clone() for l in range(L)})
    if mode in res:
        pass
    res.setdefault(mode, cur)
    if mode == 3:
        for k in cur:
            d = (cur[k] - res[1][k]).abs()
            idx = torch.nonzero(d > 1e-6)
            print(k, "maxdiff", float(d.max()), "count", idx.shape[0], "first", idx[:3].tolist())
