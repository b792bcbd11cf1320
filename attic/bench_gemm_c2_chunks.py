import os, sys
sys.path.insert(0, "nejm-brain-to-text_amd")
import torch, b2t_ops as ops
dev = torch.device("cuda:0")
ws = ops.Workspace()
def bench(name, fn, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:44s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TF/s")
H = 512
for rows in (8000, 5333, 10667, 16000):
    dG = torch.randn(rows, 4 * H, device=dev); W = torch.randn(3 * H, H, device=dev); dY = torch.empty(rows, H, device=dev)
    for sk in (1, 2, 3, 4):
        kw = dict(splitk=sk, ws=ws) if sk > 1 else {}
        bench(f"dX {rows}x512x1536 NN splitk {sk}", lambda: ops.gemm(dG, W, dY, M=rows, N_=H, K=3 * H, a_s0=4 * H, b_kc=0, b_s0=H, c_s0=H, **kw), 2.0 * rows * H * 3 * H)
for rows in (5333, 4000):
    X = torch.randn(rows, H, device=dev); Wi = torch.randn(3 * H, H, device=dev); gi = torch.empty(rows, 3 * H, device=dev)
    for sk in (1, 2):
        kw = dict(splitk=sk, ws=ws) if sk > 1 else {}
        bench(f"gi {rows}x1536x512 NT splitk {sk}", lambda: ops.gemm(X, Wi, gi, M=rows, N_=3 * H, K=H, a_s0=H, b_s0=H, c_s0=3 * H, **kw), 2.0 * rows * H * 3 * H)
