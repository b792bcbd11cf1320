"""Bimodal boxes (~19.4 vs ~20.3 ms per step from process to process): which in-process change moves a slow process?
t1 as created; t2 after re-creating the executor (new worker queues); t3 after dropping the pass workspaces (new buffers,
same executor); t4 with the device-scope hand-off.  Prints one line; run several processes per box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
import bench, b2t_ops as ops, b2t_native as N
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(10)
model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)


def run(n=25, w=4):
    for i in range(w):
        ts.step(ops.augment_smooth(x, 2, 100, "same", cut=0, white_std=1.0, offset_std=0.2, seed=i), days, labels, nts, lens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        ts.step(ops.augment_smooth(x, 2, 100, "same", cut=0, white_std=1.0, offset_std=0.2, seed=i), days, labels, nts, lens)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t1 = run()
t1b = run()
ws = model._ws
torch.cuda.synchronize()
N.load().b2t_exec_destroy(ws._exec); ws._exec = None       # next pass creates a new executor and chooses new worker queues
t2 = run()
keep = {k: v for k, v in ws.bufs.items() if k[0] == "exec_sync"}
junk = torch.empty(300 << 20, dtype=torch.uint8, device=dev)   # shifts where the re-allocated workspaces land
ws.bufs = dict(keep)
t3 = run()
ops.LOCAL_F32["dirs"] = ""
t4 = run()
print(f"as created {t1:.2f} {t1b:.2f} | new executor {t2:.2f} | new workspaces {t3:.2f} | device-scope hand-off {t4:.2f}")
