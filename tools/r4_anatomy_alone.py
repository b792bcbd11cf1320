#!/usr/bin/env python3
"""Phase anatomy (s_memtime ticks per time step, -DB2T_TIMING build) of ONE sweep alone on the chip, and of two same-class sweeps."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
os.environ.setdefault("B2T_LIB", os.path.join(ROOT, "nejm-brain-to-text_amd", "csrc", "libb2t_hip_timing.so"))
import torch
sys.argv = [sys.argv[0]] + sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("probe", os.path.join(ROOT, "tools", "r4_sweep_probe.py"))
P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
fn = ["poll", "loads+mfma", "reduce", "gates", "stagebar", "store+pub", "(split:loads)"]
bn = ["prefetch", "poll", "loads+mfma", "reduce", "gates", "stagebar", "store+pub"]
def show(tag, d):
    w = d["sync"].cpu().numpy()[8:24]
    names = fn if P.DIR == "fwd" else bn
    for blk, off in ((0, 0), (17, 8)):
        print(f"{tag} block {blk:2d}: " + " ".join(f"{n}={int(w[off + i])}" for i, n in enumerate(names)) + f" | total {int(sum(w[off:off + 7]))}")
for n, wide in ((1, False), (2, False), (1, True)):
    ds = P.mk(n)
    m = 1 | P.LOCAL | (P.WIDE if wide else 0)
    for rep in range(3):
        for d in ds: P.sweep(d, m)
        torch.cuda.synchronize()
    print(f"-- {n} sweep(s) same class, wide={wide}: " + " ".join(f"{d['e0'].elapsed_time(d['e1']) * 1e3:.0f} us" for d in ds))
    for i, d in enumerate(ds): show(f"sweep {i}", d)
