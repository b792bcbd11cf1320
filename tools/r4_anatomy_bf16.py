#!/usr/bin/env python3
"""Phase anatomy (s_memtime ticks of 10 ns per time step, -DB2T_TIMING build) of ONE bf16 32-unit sweep alone on the chip:
fragment hand-off (default) against fp32 tiles (B2T_HANDOFF16=0), forward and backward, XCD-local and device scope."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
os.environ.setdefault("B2T_LIB", os.path.join(ROOT, "nejm-brain-to-text_amd", "csrc", "libb2t_hip_timing.so"))
import torch
import importlib.util
fn = ["poll", "loads+mfma", "reduce", "gates", "stagebar", "store+pub", "(split:loads)"]
bn = ["prefetch", "poll", "loads+mfma", "reduce", "gates", "stagebar", "store+pub"]
for direction in ("fwd", "bwd"):
    os.environ["B2T_DIR"] = direction
    spec = importlib.util.spec_from_file_location("probe", os.path.join(ROOT, "tools", "r4_sweep_probe.py"))
    P = importlib.util.module_from_spec(spec); spec.loader.exec_module(P)
    for local in (P.LOCAL, 0):
        for ring in ("1", "0"):
            os.environ["B2T_HANDOFF16"] = ring
            d = P.mk(1)[0]
            m = 1 | local | P.WIDE | 0x100
            for rep in range(3):
                P.sweep(d, m)
                torch.cuda.synchronize()
            w = d["sync"].cpu().numpy()[8:24]
            names = fn if direction == "fwd" else bn
            print(f"{direction} local={int(local != 0)} ring={ring}: {d['e0'].elapsed_time(d['e1']) * 1e3:.0f} us / {P.T} steps | " +
                  " ".join(f"{n}={int(w[i])}" for i, n in enumerate(names)) + f" | total {int(sum(w[:7]))}")
