#!/usr/bin/env python3
"""Round-4 probe: what slows a backward (or forward) sweep launch inside the step?  Times each sweep of a set of
concurrent sweeps (HIP events on its own stream) alone, next to other sweeps of the same / the other XCD parity class,
and next to a stream of GEMMs (input-gradient-like: short blocks; weight-gradient-like: split-K, long blocks).
usage: r4_sweep_probe.py [T=125] ; env B2T_DIR=fwd|fused|bwd (default bwd), B2T_PROBE_MODE=<extra mode bits, hex>"""
import os, sys, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as N, b2t_ops as ops
lib = N.load()
dev = torch.device("cuda:0")
B, H = 64, 512
T = int(sys.argv[1]) if len(sys.argv) > 1 else 125
DIR = os.environ.get("B2T_DIR", "bwd")
EXTRA = int(os.environ.get("B2T_PROBE_MODE", "0"), 0)
_p = ops._p
LOCAL, PARITY, WIDE = 0x400, 0x800, 0x200


def mk(n):
    out = []
    for i in range(n):
        d = dict(gi=torch.randn(T, B, 3 * H, device=dev) * 0.1, w=torch.randn(3 * H, H, device=dev) * 0.04,
                 b=torch.zeros(3 * H, device=dev), out=torch.zeros(T + 1, B, H, device=dev),
                 res=torch.rand(T, B, 4 * H, device=dev) * 0.5, sync=torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev),
                 s=torch.cuda.Stream(), dY=torch.randn(T, B, H, device=dev) * 0.01, wt=torch.randn(H, 3 * H, device=dev) * 0.04,
                 dG=torch.empty(T, B, 4 * H, device=dev), dh=torch.empty(B, H, device=dev), sc=torch.empty(B, H, device=dev),
                 e0=torch.cuda.Event(enable_timing=True), e1=torch.cuda.Event(enable_timing=True),
                 w2=torch.randn(3 * H, H, device=dev) * 0.04, gi2=torch.empty(T, B, 3 * H, device=dev))
        out.append(d)
    return out


def sweep(d, mode):
    with torch.cuda.stream(d["s"]):
        d["e0"].record()
        if DIR == "fused":
            N.check(lib.b2t_gru_layer_fwd_fused_f32(_p(d["gi"]), _p(d["w"]), _p(d["b"]), _p(d["out"][0]), _p(d["out"][1:]), _p(d["res"]), None, _p(d["w2"]), _p(d["b"]), _p(d["gi2"]), T, B, H, mode & ~WIDE, _p(d["sync"]), ops._stream()), "ff")
        elif DIR == "fwd":
            N.check(lib.b2t_gru_layer_fwd_f32(_p(d["gi"]), _p(d["w"]), _p(d["b"]), _p(d["out"][0]), _p(d["out"][1:]), _p(d["res"]), None, T, B, H, mode, _p(d["sync"]), ops._stream()), "f")
        else:
            N.check(lib.b2t_gru_layer_bwd_f32(_p(d["dY"]), None, _p(d["res"]), _p(d["out"][1:]), _p(d["out"][0]), _p(d["wt"]), _p(d["dG"]), _p(d["dh"]), _p(d["sc"]), T, B, H, mode, _p(d["sync"]), ops._stream()), "b")
        d["e1"].record()


class Gemms:
    """A stream that keeps the chip busy with GEMMs of one kind for `n` launches."""
    def __init__(self, kind):
        self.kind = kind
        self.s = torch.cuda.Stream()
        self.ws = ops.Workspace()
        if kind == "dx":      # [8000 x 1536] x [1536 x 512]: 252 tiles, short blocks
            self.A = torch.randn(8000, 1536, device=dev); self.Bm = torch.randn(1536, 512, device=dev); self.Cm = torch.empty(8000, 512, device=dev)
        else:                 # dW-like: [1536 x 32000] x [32000 x 512], 16 K slices: 768 long blocks
            self.A = torch.randn(32000, 1536, device=dev); self.Bm = torch.randn(32000, 512, device=dev); self.Cm = torch.empty(1536, 512, device=dev)
        self.e0 = torch.cuda.Event(enable_timing=True); self.e1 = torch.cuda.Event(enable_timing=True)

    def run(self, n):
        with torch.cuda.stream(self.s):
            self.e0.record()
            for _ in range(n):
                if self.kind == "dx":
                    ops.gemm(self.A, self.Bm, self.Cm, M=8000, N_=512, K=1536, a_kc=1, b_kc=0, a_s0=1536, b_s0=512, c_s0=512)
                else:
                    ops.gemm(self.A, self.Bm, self.Cm, M=1536, N_=512, K=32000, a_kc=0, b_kc=0, a_s0=1536, b_s0=512, c_s0=512, splitk=16, ws=self.ws)
            self.e1.record()


def case(name, parities, gemm=None, ngemm=0, wide=False):
    ds = mk(len(parities))
    g = Gemms(gemm) if gemm else None
    modes = [1 | LOCAL | (PARITY if p else 0) | (WIDE if wide else 0) | EXTRA for p in parities]
    best = None
    for rep in range(4):
        torch.cuda.synchronize()
        if g: g.run(ngemm)
        for d, m in zip(ds, modes): sweep(d, m)
        torch.cuda.synchronize()
        ts = [d["e0"].elapsed_time(d["e1"]) * 1e3 for d in ds]
        tg = g.e0.elapsed_time(g.e1) * 1e3 if g else 0.0
        if rep and (best is None or sum(ts) < sum(best[0])): best = (ts, tg)
    errs = [int(d["sync"][0].item()) for d in ds]
    ts, tg = best
    print(f"{name:34s} sweeps us: {' '.join(f'{t:7.0f}' for t in ts)}  ({' '.join(f'{t / T:5.2f}' for t in ts)} us/step)"
          + (f"  gemms({gemm} x{ngemm}) {tg:7.0f} us = {tg / ngemm:6.0f} each" if g else "") + (f"  ERR {errs}" if any(errs) else ""), flush=True)


if __name__ == "__main__":
    print(f"dir={DIR} T={T} extra_mode={EXTRA:#x}")
    g = Gemms("dx"); g.run(4); torch.cuda.synchronize(); g.run(8); torch.cuda.synchronize(); print(f"dx gemm alone: {g.e0.elapsed_time(g.e1) * 1e3 / 8:6.0f} us each")
    g = Gemms("dw"); g.run(2); torch.cuda.synchronize(); g.run(4); torch.cuda.synchronize(); print(f"dw gemm alone: {g.e0.elapsed_time(g.e1) * 1e3 / 4:6.0f} us each")
    case("1 sweep", [0])
    case("2 sweeps, same class", [0, 0])
    case("2 sweeps, other class", [0, 1])
    case("3 sweeps (0,0,1)", [0, 0, 1])
    case("4 sweeps (0,0,1,1)", [0, 0, 1, 1])
    case("1 sweep + dx gemms", [0], "dx", 6)
    case("1 sweep + dw gemms", [0], "dw", 2)
    case("2 same + dx gemms", [0, 0], "dx", 6)
    case("2 other + dx gemms", [0, 1], "dx", 6)
    case("4 sweeps + dx gemms", [0, 0, 1, 1], "dx", 8)
    case("4 sweeps + dw gemms", [0, 0, 1, 1], "dw", 2)
    if DIR == "bwd":
        case("1 wide sweep", [0], wide=True)
        case("2 wide, same class", [0, 0], wide=True)
        case("4 wide (0,0,1,1)", [0, 0, 1, 1], wide=True)
        case("4 wide + dx gemms", [0, 0, 1, 1], "dx", 8, wide=True)
