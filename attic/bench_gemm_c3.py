import os, sys
sys.path.insert(0, "nejm-brain-to-text_amd")
import torch, b2t_ops as ops
dev = torch.device("cuda:0")
ws = ops.Workspace()
def bench(name, fn, flops):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:44s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TF/s")
M, N, K = 7808, 2304, 7168
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
for sk in (1, 2, 3, 4):
    if sk == 1:
        bench(f"gi0 {M}x{N}x{K} NT", lambda: ops.gemm(A, W, C, M=M, N_=N, K=K, a_s0=K, b_s0=K, c_s0=N), 2.0 * M * N * K)
    else:
        bench(f"gi0 splitk {sk}", lambda: ops.gemm(A, W, C, M=M, N_=N, K=K, a_s0=K, b_s0=K, c_s0=N, splitk=sk, ws=ws), 2.0 * M * N * K)
# dX0: dG [M x 3H(of 4H)] x W_ih0 [3H x In0] -> [M x In0]: NN with gap ignored
H3 = 2304
dG = torch.randn(M, 3072, device=dev); dV = torch.empty(M, K, device=dev)
bench("dX0 7808x7168x2304 NN", lambda: ops.gemm(dG, W, dV, M=M, N_=K, K=H3, a_s0=3072, b_kc=0, b_s0=K, c_s0=K), 2.0 * M * K * H3)
# dW_ih0: [3H x In0] = dG^T [3H x M] x A [M x In0]: TN, split-K
G = torch.empty(H3, K, device=dev)
for sk in (ops.splitk_for(H3, K, M), 2, 3, 5, 8):
    bench(f"dW_ih0 2304x7168x7808 TN splitk {sk}", lambda: ops.gemm(dG, A, G, M=H3, N_=K, K=M, a_kc=0, a_s0=3072, b_kc=0, b_s0=K, c_s0=K, splitk=sk, ws=ws), 2.0 * M * K * H3)
# layers >= 1 of the shipped shape
H = 768
X = torch.randn(M, H, device=dev); Wi = torch.randn(3 * H, H, device=dev); gi = torch.empty(M, 3 * H, device=dev)
for sk in (1, 2, 3):
    kw = dict(splitk=sk, ws=ws) if sk > 1 else {}
    bench(f"gi_l {M}x2304x768 NT splitk {sk}", lambda: ops.gemm(X, Wi, gi, M=M, N_=3 * H, K=H, a_s0=H, b_s0=H, c_s0=3 * H, **kw), 2.0 * M * H * 3 * H)
dGl = torch.randn(M, 4 * H, device=dev); dY = torch.empty(M, H, device=dev)
for sk in (1, 2, 3, 4):
    kw = dict(splitk=sk, ws=ws) if sk > 1 else {}
    bench(f"dX_l {M}x768x2304 NN splitk {sk}", lambda: ops.gemm(dGl, Wi, dY, M=M, N_=H, K=3 * H, a_s0=4 * H, b_kc=0, b_s0=H, c_s0=H, **kw), 2.0 * M * H * 3 * H)
