#!/usr/bin/env python3
"""Phase anatomy of the persistent forward sweep (-DB2T_TIMING build, csrc/libb2t_hip_timing.so) at 1, 2 and 4
workgroups per CU (CU-masked streams).  Numbers are s_memtime ticks per step, as the kernel reports them.
Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DB2T_TIMING -x hip -c gru_persistent.hip -o t.o  and link it
with the other objects into csrc/libb2t_hip_timing.so."""
import os, sys, ctypes as C, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import b2t_native as N
N.LIB_PATH = os.path.join(ROOT, "nejm-brain-to-text_amd", "csrc", os.environ.get("B2T_TIMING_LIB", "libb2t_hip_timing.so"))
import b2t_ops as ops
lib = N.load()
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0"); torch.zeros(1, device=dev)
B, H, T = 64, 512, 250
_p = ops._p
def masked_stream(slices):
    m = (C.c_uint32 * 8)()
    for k in slices: m[k] = 0xffffffff
    s = C.c_void_p(); assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, m) == 0
    return torch.cuda.ExternalStream(s.value)
def mk(stream):
    return dict(gi=torch.randn(T, B, 3 * H, device=dev) * 0.1, w=torch.randn(3 * H, H, device=dev) * 0.04, b=torch.zeros(3 * H, device=dev),
                out=torch.zeros(T + 1, B, H, device=dev), res=torch.empty(T, B, 4 * H, device=dev),
                sync=torch.zeros(lib.b2t_gru_ws_bytes(T, B, H) // 4 + 16, dtype=torch.int32, device=dev), s=stream)
def fwd(d):
    with torch.cuda.stream(d["s"]):
        N.check(lib.b2t_gru_layer_fwd_f32(_p(d["gi"]), _p(d["w"]), _p(d["b"]), _p(d["out"][0]), _p(d["out"][1:]), _p(d["res"]), None, T, B, H, MODE, _p(d["sync"]), ops._stream()), "f")
MODE = int(os.environ.get("B2T_GRU_MODE", "1"))
names = ["poll", "loads+mfma", "reduce", "gates", "stagebar", "store+pub", "(split:loads)"] if MODE == 1 else \
        ["drain", "repoll", "issue", "mfma", "reduce", "gates", "stagebar", "store+pub"]
for tag, plans in (("1 WG/CU", [[0, 1, 2, 3]]), ("2 WG/CU", [[0, 1]]), ("4 WG/CU", [[0]]), ("2 sweeps on 4 slices", [[0, 1, 2, 3]] * 2)):
    ds = [mk(masked_stream(p)) for p in plans]
    for rep in range(3):
        t0 = time.perf_counter()
        for d in ds: fwd(d)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    w = ds[0]["sync"][8:24].cpu().tolist()
    print(f"{tag:22s} {dt/T*1e6:6.2f} us/step | block0: " + " ".join(f"{n}={w[i]}" for i, n in enumerate(names)) + " | block17: " + " ".join(f"{n}={w[8+i]}" for i, n in enumerate(names)))
