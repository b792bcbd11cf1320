#!/usr/bin/env python3
"""Persistent GRU sweeps on CU-masked streams (hipExtStreamCreateWithCUMask): per-CU co-residency scaling and
partition plans.  Mask bit i -> XCD i%8, SE (i/8)%4, CU #(i/32) of that SE (measured, tools/ubench/placement.hip),
so bits [32k, 32k+32) are one 'slice' = CU #k of every SE of every XCD."""
import os, sys, ctypes as C, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as N, b2t_ops as ops
lib = N.load()
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
B, H = 64, 512
T = int(sys.argv[1]) if len(sys.argv) > 1 else 250
_p = ops._p

def masked_stream(slices):
    m = (C.c_uint32 * 8)()
    for k in slices:
        m[k] = 0xffffffff
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, m)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def mk(stream):
    return dict(gi=torch.randn(T, B, 3 * H, device=dev) * 0.1, w=torch.randn(3 * H, H, device=dev) * 0.04,
                b=torch.zeros(3 * H, device=dev), out=torch.zeros(T + 1, B, H, device=dev),
                res=torch.empty(T, B, 4 * H, device=dev),
                sync=torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev),
                s=stream, dY=torch.randn(T, B, H, device=dev) * 0.01, wt=torch.randn(H, 3 * H, device=dev) * 0.04,
                dG=torch.empty(T, B, 4 * H, device=dev), dh=torch.empty(B, H, device=dev), sc=torch.empty(B, H, device=dev))
def fwd(d):
    with torch.cuda.stream(d["s"]):
        N.check(lib.b2t_gru_layer_fwd_f32(_p(d["gi"]), _p(d["w"]), _p(d["b"]), _p(d["out"][0]), _p(d["out"][1:]), _p(d["res"]), None, T, B, H, 1, _p(d["sync"]), ops._stream()), "f")
def bwd(d):
    with torch.cuda.stream(d["s"]):
        N.check(lib.b2t_gru_layer_bwd_f32(_p(d["dY"]), None, _p(d["res"]), _p(d["out"][1:]), _p(d["out"][0]), _p(d["wt"]), _p(d["dG"]), _p(d["dh"]), _p(d["sc"]), T, B, H, 1, _p(d["sync"]), ops._stream()), "b")

def run(tag, plans):
    """plans: list of slice lists, one per concurrent sweep."""
    ds = [mk(masked_stream(p)) for p in plans]
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for d in ds: fwd(d)
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            for d in ds: fn(d)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        n = len(ds)
        print(f"{tag:34s} {name}: wall {dt*1e3:7.3f} ms -> {dt/T*1e6:6.2f} us/step, {dt/T/n*1e6:6.2f} us per sweep-step", flush=True)

run("1 sweep, 4 slices (1 WG/CU)", [[0, 1, 2, 3]])
run("1 sweep, 8 slices (spread)", [list(range(8))])
run("1 sweep, 2 slices (2 WG/CU)", [[0, 1]])
run("1 sweep, 1 slice  (4 WG/CU)", [[0]])
run("2 sweeps, disjoint 4+4", [[0, 1, 2, 3], [4, 5, 6, 7]])
run("2 sweeps, same 4 slices", [[0, 1, 2, 3], [0, 1, 2, 3]])
run("4 sweeps, disjoint 2 each", [[0, 1], [2, 3], [4, 5], [6, 7]])
run("4 sweeps, pairs share 4", [[0, 1, 2, 3], [0, 1, 2, 3], [4, 5, 6, 7], [4, 5, 6, 7]])
run("5 sweeps, all 8 slices", [list(range(8))] * 5)
run("5 sweeps, 2,2,2,1,1", [[0, 1], [2, 3], [4, 5], [6], [7]])
run("3 sweeps, 3,3,2", [[0, 1, 2], [3, 4, 5], [6, 7]])
