#!/usr/bin/env python3
"""Concurrency scaling of the persistent GRU sweeps: N independent layer sweeps on N streams."""
import os, sys, ctypes as C
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as N, b2t_ops as ops
lib = N.load()
dev = torch.device("cuda:0")
B, H = int(os.environ.get("B2T_B", "64")), 512
T = int(sys.argv[1]) if len(sys.argv) > 1 else 125
FMODE = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_p = ops._p
def mk(n):
    out = []
    for i in range(n):
        d = dict(gi=torch.randn(T, B, 3 * H, device=dev) * 0.1, w=torch.randn(3 * H, H, device=dev) * 0.04,
                 b=torch.zeros(3 * H, device=dev), out=torch.zeros(T + 1, B, H, device=dev),
                 res=torch.empty(T, B, 4 * H, device=dev), sync=torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev),
                 s=torch.cuda.Stream(), dY=torch.randn(T, B, H, device=dev) * 0.01, wt=torch.randn(H, 3 * H, device=dev) * 0.04,
                 dG=torch.empty(T, B, 4 * H, device=dev), dh=torch.empty(B, H, device=dev), sc=torch.empty(B, H, device=dev))
        d["xoff"] = (3 * i) % 8
        out.append(d)
    return out
def mode_of(d, m):
    return m
def fwd(d, m=None):
    with torch.cuda.stream(d["s"]):
        N.check(lib.b2t_gru_layer_fwd_f32(_p(d["gi"]), _p(d["w"]), _p(d["b"]), _p(d["out"][0]), _p(d["out"][1:]), _p(d["res"]), None, T, B, H, mode_of(d, FMODE if m is None else m), _p(d["sync"]), ops._stream()), "f")
def bwd(d, m=None):
    with torch.cuda.stream(d["s"]):
        N.check(lib.b2t_gru_layer_bwd_f32(_p(d["dY"]), None, _p(d["res"]), _p(d["out"][1:]), _p(d["out"][0]), _p(d["wt"]), _p(d["dG"]), _p(d["dh"]), _p(d["sc"]), T, B, H, mode_of(d, (FMODE if FMODE != 2 else 1) if m is None else m), _p(d["sync"]), ops._stream()), "b")
import time
# parity of the chosen mode against the step-launch kernels (mode 0) on the same data
d = mk(2)[1]
fwd(d, 0); torch.cuda.synchronize(); o0, r0 = d["out"].clone(), d["res"].clone()
bwd(d, 0); torch.cuda.synchronize(); g0, h0 = d["dG"].clone(), d["dh"].clone()
d["out"][1:].zero_(); d["res"].zero_(); d["dG"].zero_(); d["dh"].zero_()
fwd(d); torch.cuda.synchronize()
print("fwd parity vs mode 0: out", float((d["out"] - o0).abs().max()), "reserve", float((d["res"] - r0).abs().max()))
d["out"].copy_(o0); d["res"].copy_(r0)
bwd(d); torch.cuda.synchronize()
print("bwd parity vs mode 0: dG", float((d["dG"] - g0).abs().max()), "dh0", float((d["dh"] - h0).abs().max()), "| max|dG|", float(g0.abs().max()))
err = int(d["sync"][0].item()); print("error word:", err)
NS = [int(x) for x in os.environ.get("B2T_NS", "1,2,3,4,5,6").split(",")]
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    if os.environ.get("B2T_DIR", name) != name: continue
    for n in NS:
        ds = mk(n)
        for d in ds: fwd(d)
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            for d in ds: fn(d)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        errs = [int(d["sync"][0].item()) for d in ds]
        print(f"{name} N={n}: errs {errs} wall {dt*1e3:7.3f} ms  -> {dt/T*1e6:6.2f} us/step wall, {dt/T/n*1e6:6.2f} us per sweep-step")

