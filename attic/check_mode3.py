#!/usr/bin/env python3
"""Mode 3 (pipelined) vs mode 1 sweeps on identical inputs, repeated: forward out/reserve and backward dG/dh0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as N, b2t_ops as ops
lib = N.load(); dev = torch.device("cuda:0"); _p = ops._p
B, H, T = int(sys.argv[1]) if len(sys.argv) > 1 else 40, 512, int(sys.argv[2]) if len(sys.argv) > 2 else 60
torch.manual_seed(1)
gi = torch.randn(T, B, 3 * H, device=dev) * 0.5; w = torch.randn(3 * H, H, device=dev) * 0.05; b = torch.randn(3 * H, device=dev) * 0.1
h0 = torch.randn(B, H, device=dev) * 0.1
dY = torch.randn(T, B, H, device=dev) * 0.01; wt = w.t().contiguous()
NCHUNK = int(os.environ.get("NCHUNK", "1"))
SYNC = torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev)
import ctypes as C
def run(mode, junk):
    out = torch.full((T + 1, B, H), junk, device=dev); out[0] = h0
    res = torch.full((T, B, 4 * H), junk, device=dev)
    sync = SYNC if NCHUNK > 1 else torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev)
    n = T // NCHUNK
    for c in range(NCHUNK):
        t0 = c * n
        N.check(lib.b2t_gru_layer_fwd_f32(C.c_void_p(gi.data_ptr() + 4 * t0 * B * 3 * H), _p(w), _p(b), C.c_void_p(out.data_ptr() + 4 * t0 * B * H),
                                          C.c_void_p(out.data_ptr() + 4 * (1 + t0) * B * H), C.c_void_p(res.data_ptr() + 4 * t0 * B * 4 * H), None, n, B, H, mode, _p(sync), ops._stream()), "f")
    dG = torch.full((T, B, 4 * H), junk, device=dev); dh = torch.full((B, H), junk, device=dev); sc = torch.empty(B, H, device=dev)
    N.check(lib.b2t_gru_layer_bwd_f32(_p(dY), None, _p(res), _p(out[1:]), _p(out[0]), _p(wt), _p(dG), _p(dh), _p(sc), T, B, H, mode, _p(sync), ops._stream()), "b")
    torch.cuda.synchronize()
    return out, res, dG, dh, int(sync[0])
ref = run(1, 0.0)
bad = 0
for rep in range(40):
    got = run(3, float(rep % 3) * 0.37)
    d = [float((a - r).abs().max()) for a, r in zip(got[:4], ref[:4])]
    n = [int(((a - r).abs() > 1e-6).sum()) for a, r in zip(got[:4], ref[:4])]
    if max(d) > 1e-6 or got[4]:
        bad += 1
        if bad <= 6: print(f"rep {rep}: maxdiff out/res/dG/dh {d} counts {n} err {got[4]}")
print("bad reps:", bad, "of 40")
