#!/usr/bin/env python3
"""Round-5 probe of the paired backward sweep (W_hh^T in LDS, tools/r4_sweep_probe.py's cases): one sweep alone, two / four in
flight on different XCD sets, next to dx-like and dW-like GEMM streams -- beside the register-resident sweep's numbers from the same
process.  usage: r5_pair_probe.py [T=125]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import r4_sweep_probe as P
PAIRED, SHIFT = 0x1000, 13


def case(name, sets, gemm=None, ngemm=0, paired=True):
    ds = P.mk(len(sets))
    g = P.Gemms(gemm) if gemm else None
    modes = [(1 | P.LOCAL | PAIRED | (s << SHIFT)) if paired else (1 | P.LOCAL | (P.PARITY if s & 1 else 0)) for s in sets]
    best = None
    for rep in range(4):
        torch.cuda.synchronize()
        if g: g.run(ngemm)
        for d, m in zip(ds, modes): P.sweep(d, m)
        torch.cuda.synchronize()
        ts = [d["e0"].elapsed_time(d["e1"]) * 1e3 for d in ds]
        tg = g.e0.elapsed_time(g.e1) * 1e3 if g else 0.0
        if rep and (best is None or sum(ts) < sum(best[0])): best = (ts, tg)
    errs = [int(d["sync"][0].item()) for d in ds]
    ts, tg = best
    print(f"{('paired ' if paired else 'regs   ') + name:40s} sweeps us: {' '.join(f'{t:7.0f}' for t in ts)}  ({' '.join(f'{t / P.T:5.2f}' for t in ts)} us/step)"
          + (f"  gemms({gemm} x{ngemm}) {tg:7.0f} us = {tg / ngemm:6.0f} each" if g else "") + (f"  ERR {errs}" if any(errs) else ""), flush=True)


if __name__ == "__main__":
    g = P.Gemms("dx"); g.run(4); torch.cuda.synchronize(); g.run(8); torch.cuda.synchronize(); print(f"dx gemm alone: {g.e0.elapsed_time(g.e1) * 1e3 / 8:6.0f} us each")
    g = P.Gemms("dw"); g.run(2); torch.cuda.synchronize(); g.run(4); torch.cuda.synchronize(); print(f"dw gemm alone: {g.e0.elapsed_time(g.e1) * 1e3 / 4:6.0f} us each")
    for paired in (True, False):
        case("1 sweep", [0], paired=paired)
        case("2 sweeps (sets 0,1)", [0, 1], paired=paired)
        case("4 sweeps (sets 0,1,2,3)", [0, 1, 2, 3], paired=paired)
        case("1 sweep + dx gemms", [0], "dx", 6, paired=paired)
        case("1 sweep + dw gemms", [0], "dw", 2, paired=paired)
        case("4 sweeps + dx gemms", [0, 1, 2, 3], "dx", 8, paired=paired)
        case("4 sweeps + dw gemms", [0, 1, 2, 3], "dw", 2, paired=paired)
        case("4 sweeps + dw gemms x4", [0, 1, 2, 3], "dw", 4, paired=paired)
