#!/usr/bin/env python3
"""Round 5: the packed bf16 GEMM's 256 x 256 kernel against the 128 x 128 one (B2T_GEMM_256 = 0 / 2, read per call) on the step's
chip-filling products: results must be bit-identical (same k order per output element); times are whole calls incl. the two packs
(identical in both modes: the difference is the GEMM kernels'), and -- under rocprofv3 --kernel-trace -- tools/r5_gemm256_trace.py
reads the kernels' own durations from the trace."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as N, b2t_ops as ops
lib = N.load(); dev = torch.device("cuda:0")
SHAPES = (("dW_ih0 2304x7168x7808", 2304, 7168, 7808), ("dX0 3904x7168x2304", 3904, 7168, 2304), ("dX0 whole 7808x7168x2304", 7808, 7168, 2304),
          ("gi0 whole 7808x2304x7168", 7808, 2304, 7168), ("gi0 chunk 2624x2304x7168", 2624, 2304, 7168), ("4096^3", 4096, 4096, 4096),
          ("odd 1000x3000x520", 1000, 3000, 520), ("gi l>=1 chunk 2624x2304x768", 2624, 2304, 768), ("dX l>=1 chunk 3904x768x2304", 3904, 768, 2304),
          ("c2 gi chunk 5376x1536x512", 5376, 1536, 512), ("c2 dX chunk 8000x512x1536", 8000, 512, 1536), ("c2 dW_ih 1536x512x32000 splitk 16", 1536, 512, 32000))
bad = 0
for name, M, Nn, K in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(dev); B = torch.randn(Nn, K, generator=g).to(dev)
    wsb = lib.b2t_gemm_bf16p_ws_bytes(M, Nn, K)
    ws = torch.empty(wsb // 4 + 64, dtype=torch.float32, device=dev)
    res = {}
    for mode in ("0", "2"):
        os.environ["B2T_GEMM_256"] = mode
        Cm = torch.full((M, Nn), float("nan"), device=dev)
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
        res[mode] = (dt, Cm)
    same = torch.equal(res["0"][1], res["2"][1])
    ref = (A.bfloat16().float() @ B.bfloat16().float().t())
    err = float((res["2"][1] - ref).abs().max() / ref.abs().max())
    bad += (not same) or not (err < 1e-4)
    print(f"{name}: 128-tile {res['0'][0] * 1e6:.0f} us, 256-tile {res['2'][0] * 1e6:.0f} us incl. packs; bit-identical {same}; max rel err vs torch {err:.1e}", flush=True)
os.environ.pop("B2T_GEMM_256", None)
print("RESULT:", "clean" if not bad else "MISMATCH")
sys.exit(1 if bad else 0)
