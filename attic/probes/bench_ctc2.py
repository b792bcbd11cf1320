#!/usr/bin/env python3
"""CTC loss + gradient at the C2 shape (B=64, T=500, S<=60): wave-synchronous recursion against thread-per-state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_ops as ops
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
B, T, C, S = 64, 500, 41, 60
logits = torch.from_numpy((rng.standard_normal((B, T, C))).astype(np.float32)).to(dev)
tg = torch.from_numpy(rng.integers(1, C, (B, S)).astype(np.int32)); tl = torch.from_numpy(rng.integers(20, S + 1, B).astype(np.int32))
il = torch.full((B,), T, dtype=torch.int32)
ws = ops.Workspace()
for mode in ("0", "1", "0", "1"):
    os.environ["B2T_CTC_WAVE"] = mode
    for _ in range(5):
        ops.ctc_loss(logits, tg, il, tl, True, 1.0 / B, ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        loss, dl, _ = ops.ctc_loss(logits, tg, il, tl, True, 1.0 / B, ws)
    e1.record(); torch.cuda.synchronize()
    print(f"B2T_CTC_WAVE={mode}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (recursions + gradient), loss[0] {float(loss[0]):.4f}")
