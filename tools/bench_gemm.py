#!/usr/bin/env python3
"""Time every GEMM shape of the C2 training step in isolation (HIP events), print TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_ops as ops

dev = torch.device("cuda:0")
B, T, F, H, C = 64, 500, 512, 512, 41
M = B * T
ws = ops.Workspace()
big = lambda *s: torch.randn(*s, device=dev)
x = big(B, T, F); W = big(45, F, F); U = big(B, T, F); bias = big(45, 1024)
day = torch.tensor([0, 11, 22, 33]).repeat_interleave(16).to(dev, torch.int32)
wih = big(3 * H, H); gi = big(T, B, 3 * H); out = big(T + 1, B, H); dG = big(T, B, 4 * H)
logits = big(B, T, C); dl = big(B, T, 48); wout = big(C, H); dY = big(T, B, H); gw = big(3 * H, H); slab = big(B, F, F)
gout = big(C, H)

cases = {
    "day_fwd Z64 500x512x512": (lambda: ops.gemm(x, W, U, M=T, N_=F, K=F, Z=B, a_kc=1, a_s0=F, a_sz=T * F, b_kc=0, b_s0=F, b_sz=F * F, c_s0=F, c_sz=T * F, bias=bias, bias_sz=1024, b_zmap=day, epilogue=1), 2.0 * B * T * F * F),
    "gi_l0 Z64 500x1536x512 NT": (lambda: ops.gemm(U, wih, gi, M=T, N_=3 * H, K=F, Z=B, a_kc=1, a_s0=F, a_sz=T * F, b_kc=1, b_s0=F, c_s0=B * 3 * H, c_sz=3 * H), 2.0 * M * 3 * H * F),
    "gi_l1 32000x1536x512 NT": (lambda: ops.gemm(out[1:], wih, gi, M=M, N_=3 * H, K=H, a_kc=1, a_s0=H, b_kc=1, b_s0=H, c_s0=3 * H), 2.0 * M * 3 * H * H),
    "head 32000x41x512 NT": (lambda: ops.gemm(out[1:], wout, logits, M=M, N_=C, K=H, a_kc=1, a_s0=H, b_kc=1, b_s0=H, c_div=B, c_s1=C, c_s0=T * C), 2.0 * M * C * H),
    "head_bwd dY 32000x512x41 NN": (lambda: ops.gemm(dl, wout, dY, M=M, N_=H, K=C, a_kc=1, a_div=B, a_s1=48, a_s0=T * 48, b_kc=0, b_s0=H, c_s0=H), 2.0 * M * C * H),
    "head_bwd dW 41x512x32000 TN": (lambda: ops.gemm(dl, out[1:], gout, M=C, N_=H, K=M, a_kc=0, a_div=B, a_s1=48, a_s0=T * 48, b_kc=0, b_s0=H, c_s0=H), 2.0 * M * C * H),
    "dX 32000x512x1024 NN": (lambda: ops.gemm(dG, wih, dY, M=M, N_=H, K=2 * H, a_kc=1, a_s0=4 * H, b_kc=0, b_s0=H, c_s0=H), 2.0 * M * H * 2 * H),
    "dX2 32000x512x512 NN acc": (lambda: ops.gemm(dG, wih, dY, M=M, N_=H, K=H, a_kc=1, a_s0=4 * H, a_off=3 * H, b_kc=0, b_s0=H, b_off=2 * H * H, c_s0=H, accumulate=1), 2.0 * M * H * H),
    "dW_hh 1536x512x32000 TN sk": (lambda: ops.gemm(dG, out, gw, M=3 * H, N_=H, K=M, a_kc=0, a_s0=4 * H, b_kc=0, b_s0=H, c_s0=H, splitk=ops.splitk_for(3 * H, H, M), ws=ws), 2.0 * M * 3 * H * H),
    "dW_hh 1536x512x32000 TN nosplit": (lambda: ops.gemm(dG, out, gw, M=3 * H, N_=H, K=M, a_kc=0, a_s0=4 * H, b_kc=0, b_s0=H, c_s0=H), 2.0 * M * 3 * H * H),
    "dW_ih0 1024x512x32000 TN sk (b_div)": (lambda: ops.gemm(dG, U, gw, M=2 * H, N_=F, K=M, a_kc=0, a_s0=4 * H, b_kc=0, b_div=B, b_s1=F, b_s0=T * F, c_s0=F, splitk=ops.splitk_for(2 * H, F, M), ws=ws), 2.0 * M * 2 * H * F),
    "day_slab Z64 512x512x500 TN": (lambda: ops.gemm(x, U, slab, M=F, N_=F, K=T, Z=B, a_kc=0, a_s0=F, a_sz=T * F, b_kc=0, b_s0=F, b_sz=T * F, c_s0=F, c_sz=F * F), 2.0 * B * T * F * F),
}
for name, (fn, flops) in cases.items():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:40s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s")

# calibration: large square problems (the guide's untuned 128x128 f32-MFMA kernel reaches 122 TF at 4096^3)
for n in (2048, 4096):
    A = torch.randn(n, n, device=dev); Bm = torch.randn(n, n, device=dev); Cm = torch.empty(n, n, device=dev)
    for kc in ((1, 1), (1, 0), (0, 0)):
        fn = lambda: ops.gemm(A, Bm, Cm, M=n, N_=n, K=n, a_kc=kc[0], b_kc=kc[1], a_s0=n, b_s0=n, c_s0=n)
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [fn() for _ in range(5)]; e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"square {n}^3 akc={kc[0]} bkc={kc[1]}: {ms*1e3:9.1f} us  {2.0*n**3/ms/1e9:7.1f} TFLOP/s")
