#!/usr/bin/env python3
"""Phase anatomy of the forward sweeps INSIDE the training step (B2T_LIB = a -DB2T_TIMING build, tools/build_timing_lib.sh):
s_memtime ticks per time step of the last chunk of every layer, workgroups 0 and 17 of row group 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
os.environ.setdefault("B2T_LIB", os.path.join(ROOT, "nejm-brain-to-text_amd", "csrc", "libb2t_hip_timing.so"))
import numpy as np, torch
import bench
import b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(10)
model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)
for i in range(6):
    feats = ops.augment_smooth(x, 2, 100, "same", cut=0, white_std=1.0, offset_std=0.2, seed=i)
    ts.step(feats, days, labels, nts, lens)
torch.cuda.synchronize()
sync = model._ws.sync(bench.L, dev).cpu().numpy()
names = ["poll", "loads+mfma", "reduce", "gates", "stagebar", "store+pub", "(split:loads)"]
bnames = ["prefetch", "poll", "loads+mfma", "reduce", "gates", "stagebar", "store+pub"]
for l in range(bench.L):
    w = sync[bench.L + l][8:24]
    for blk, off in ((0, 0), (17, 8)):
        tot = int(sum(w[off:off + 7]))
        print(f"bwd layer {l} block {blk:2d}: " + " ".join(f"{n}={int(w[off + i])}" for i, n in enumerate(bnames)) + f" | total {tot}")
for l in range(bench.L):
    w = sync[l][8:24]
    for blk, off in ((0, 0), (17, 8)):
        tot = int(sum(w[off:off + 6]))
        print(f"layer {l} block {blk:2d}: " + " ".join(f"{n}={int(w[off + i])}" for i, n in enumerate(names[:6])) + f" | total {tot}")

# placement of the LAST backward launch (layer 0, first time chunk): per-workgroup start / first hand-off / end, 100 MHz wall clock
import ctypes
lib = ctypes.CDLL(os.environ["B2T_LIB"])
if hasattr(lib, "b2t_debug_bwd_times"):
    buf = (ctypes.c_uint32 * (3 * 512))()
    lib.b2t_debug_bwd_times(buf)
    a = np.array(buf, dtype=np.int64).reshape(3, 512)
    n = int((a[0] != 0).sum())
    st, first, en = a[0][:n], a[1][:n], a[2][:n]
    t0 = st.min()
    print(f"last backward launch: {n} workgroups; start spread {(st.max() - t0) / 100:.1f} us; first hand-off received at "
          f"{(first.min() - t0) / 100:.1f} .. {(first.max() - t0) / 100:.1f} us; end {(en.max() - t0) / 100:.1f} us after the first workgroup started")
    print("  start times by row group (us):", [f"{(st[g * 32:(g + 1) * 32].min() - t0) / 100:.0f}-{(st[g * 32:(g + 1) * 32].max() - t0) / 100:.0f}" for g in range(n // 32)])
