"""Why do some boxes alternate between ~20.0 and ~21.5 ms per step from process to process?  Fingerprint a process:
device pointers of the big buffers, calibration hops, clocks, then the step time measured twice in the same process."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import bench, b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(10)
model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)
def run(n):
    for i in range(5):
        ts.step(ops.augment_smooth(x, 2, 100, "same", cut=0, white_std=1.0, offset_std=0.2, seed=i), days, labels, nts, lens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        ts.step(ops.augment_smooth(x, 2, 100, "same", cut=0, white_std=1.0, offset_std=0.2, seed=i), days, labels, nts, lens)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
a = run(30); b = run(30)
ws = [v for k, v in model._ws.__dict__.items() if isinstance(v, dict)]
ptrs = []
for d in ws:
    for k, t in d.items():
        if hasattr(t, "data_ptr") and t.numel() > 1 << 20:
            ptrs.append(hex(t.data_ptr()))
try:
    clk = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    clk = " ".join(l.split(":")[-1].strip() for l in clk.splitlines() if "sclk" in l or "mclk" in l)
except Exception as e:
    clk = "n/a"
print(f"{a:.2f} {b:.2f} ms | x {hex(x.data_ptr())} params {hex(model.arena().data_ptr())} ws {ptrs[:3]} | {clk}")
