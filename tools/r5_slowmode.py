#!/usr/bin/env python3
"""Round-5 verdict item 7: what differs between a fast and a 'slow-mode' process?  One process = one line of JSON: host latency of
the runtime calls the plan is made of (a small kernel launch through the C ABI, hipEventRecord, hipStreamWaitEvent: p50 / p90 of 400
calls each, taken first thing in the process), the executor's measured cross-queue hop times (B2T_PLAN_DUMP, stderr), then the
headline step for 12 steps (ms per step, host enqueue per step).  Run many times in one gpurun call with different environments
(tools/run_r5e.sh): HSA_ENABLE_INTERRUPT=0, GPU_MAX_HW_QUEUES=4/8, AMD_DIRECT_DISPATCH=0, B2T_WORKERS=1."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np
import torch
import bench, b2t_native as N, b2t_ops as ops
from rnn_model import GRUDecoder
from b2t_train_step import TrainStep

dev = torch.device("cuda:0")
lib = N.load()
torch.cuda.synchronize()


def pct(v):
    v = np.array(v) * 1e6
    return dict(p50=round(float(np.percentile(v, 50)), 2), p90=round(float(np.percentile(v, 90)), 2), max=round(float(v.max()), 1))


out = dict(tag=os.environ.get("R5_TAG", "default"), pid=os.getpid())
a = torch.zeros(64, 64, device=dev); b = torch.zeros(64, 64, device=dev)
s2 = torch.cuda.Stream()
ev = torch.cuda.Event()
for name, fn in (("kernel_launch", lambda: lib.b2t_transpose_f32(ops._p(a), ops._p(b), 64, 64, ops._stream())),
                 ("event_record", lambda: ev.record()),
                 ("stream_wait_event", lambda: s2.wait_event(ev))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    ts = []
    for i in range(400):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        if i % 50 == 49:
            torch.cuda.synchronize()
    out[name + "_us"] = pct(ts)
torch.manual_seed(10)
model = GRUDecoder(bench.F, bench.H, bench.D, bench.C, 0.0, 0.0, bench.L, 0, 0).to(dev).train()
ts_ = TrainStep(model, dict(bench.ARGS))
x, days, labels, nts, lens = bench.make_batch(1000, dev)
def step(i):
    return ts_.step(ops.augment_smooth(x, 2, 100, "same", cut=i % 3, white_std=1.0, offset_std=0.2, seed=i), days, labels, nts - i % 3, lens)
for i in range(5): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(12): step(5 + i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); dt = time.perf_counter() - t0
ts_.check_status()
out.update(ms_per_step=round(dt / 12 * 1e3, 3), host_enqueue_ms_per_step=round(t_enq / 12 * 1e3, 3))
try:
    out["cpu"] = dict(affinity=len(os.sched_getaffinity(0)), on=int(open(f"/proc/{os.getpid()}/stat").read().split()[38]))
except Exception:
    pass
print("R5SLOW " + json.dumps(out), flush=True)
