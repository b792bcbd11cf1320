#!/usr/bin/env python3
"""Full training step, mode 3 vs mode 1 gradients on identical data (pipelined plan), for several batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
dev = torch.device("cuda:0")
F, H, D, C, L, T = 512, 512, 4, 41, int(os.environ.get("L", "2")), int(os.environ.get("T", "60"))
args = dict(lr_max=0.005, lr_min=0.0001, lr_decay_steps=1000, lr_warmup_steps=10, lr_scheduler_type="cosine", lr_max_day=0.005,
            lr_min_day=0.0001, lr_decay_steps_day=1000, lr_warmup_steps_day=10, beta0=0.9, beta1=0.999, epsilon=0.1,
            weight_decay=0.001, weight_decay_day=0, grad_norm_clip_value=10, _debug_keep_unclipped=True)
for B in [int(x) for x in os.environ.get("BS", "40,64").split(",")]:
    torch.manual_seed(3)
    x = torch.randn(B, T, F, device=dev); day = torch.randint(0, D, (B,), device=dev, dtype=torch.int32)
    tgt = torch.randint(1, C, (B, 12), device=dev, dtype=torch.int32); nt = torch.full((B,), T, device=dev, dtype=torch.int32)
    tl = torch.full((B,), 12, device=dev, dtype=torch.int32)
    res = {}
    for mode in (1, 3):
        ops.GRU_MODE["value"] = mode
        torch.manual_seed(5)
        model = GRUDecoder(F, H, D, C, 0.0, 0.0, L, 0, 0).to(dev).train()
        ts = TrainStep(model, dict(args))
        worst = []
        for rep in range(int(os.environ.get("REPS", "6"))):
            ts.step(x, day, tgt, nt, tl)
            torch.cuda.synchronize()
            worst.append({k: v.copy() for k, v in ts.last_unclipped_grads().items()})
        res[mode] = worst
    bad = 0
    for rep, (g1, g3) in enumerate(zip(res[1], res[3])):
        for k in g1:
            d = np.abs(g1[k] - g3[k]).max(); sc = max(1e-9, np.abs(g1[k]).max())
            if d > 1e-5 * sc + 1e-9:
                bad += 1
                if bad < 8: print(f"B={B} rep {rep} {k}: maxdiff {d:.3e} (scale {sc:.3e}), n>{int((np.abs(g1[k]-g3[k]) > 1e-5*sc).sum())}")
    print(f"B={B}: mismatching tensors {bad}")
