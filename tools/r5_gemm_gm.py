#!/usr/bin/env python3
"""Round 5: the packed bf16 GEMM on the shipped shape's layer-0 products (wide: 56 tile columns) under the row-major and the grouped
tile order (B2T_GEMM_GM, read once per process: run twice).  Prints TF/s per shape."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as N, b2t_ops as ops
lib = N.load(); dev = torch.device("cuda:0")
out = []
for name, M, Nn, K in (("gi0 chunk (2624 x 2304 x 7168)", 2624, 2304, 7168), ("gi0 whole (7808 x 2304 x 7168)", 7808, 2304, 7168),
                       ("dX0 (3904 x 7168 x 2304)", 3904, 7168, 2304), ("dW_ih0 (2304 x 7168 x 7808)", 2304, 7168, 7808),
                       ("gi l>=1 (2624 x 2304 x 768)", 2624, 2304, 768), ("c2 gi (5376 x 1536 x 512)", 5376, 1536, 512)):
    A = torch.randn(M, K, device=dev); B = torch.randn(Nn, K, device=dev); Cm = torch.empty(M, Nn, device=dev)
    wsb = lib.b2t_gemm_bf16p_ws_bytes(M, Nn, K)
    ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
    d = N.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cm.data_ptr()
    d.M, d.N, d.K, d.Z = M, Nn, K, 1
    d.a_kcontig, d.b_kcontig, d.a_s0, d.b_s0, d.c_s0 = 1, 1, K, K, Nn
    d.splitk = 1
    for _ in range(3):
        N.check(lib.b2t_gemm_bf16p_f32(C.byref(d), ops._p(ws), wsb, ops._stream()), "gemm")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        N.check(lib.b2t_gemm_bf16p_f32(C.byref(d), ops._p(ws), wsb, ops._stream()), "gemm")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    out.append(f"{name}: {dt * 1e6:.0f} us incl. packs = {2.0 * M * Nn * K / dt / 1e12:.0f} TF/s")
print(f"B2T_GEMM_GM={os.environ.get('B2T_GEMM_GM', '(default 8)')}: " + " | ".join(out))
