#!/usr/bin/env python3
"""CTC kernel timing at the bench shape: loss only (alpha pass) vs loss + gradient (alpha + beta)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_ops as ops
dev = torch.device("cuda:0")
B, T, C, S = 64, 500, 41, 60
torch.manual_seed(0)
logits = torch.randn(B, T, C, device=dev)
targets = torch.randint(1, C, (B, S), device=dev, dtype=torch.int32)
il = torch.full((B,), T, dtype=torch.int32, device=dev); tl = torch.full((B,), S, dtype=torch.int32, device=dev)
ws = ops.Workspace()
for want in (False, True):
    for _ in range(3): ops.ctc_loss(logits, targets, il, tl, want, 1.0 / B, ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.ctc_loss(logits, targets, il, tl, want, 1.0 / B, ws)
    e1.record(); torch.cuda.synchronize()
    print("grad" if want else "loss only", f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
