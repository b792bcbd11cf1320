import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, "/root/repo/nejm-brain-to-text_amd")
import torch, b2t_ops as ops, b2t_native as N
dev = torch.device("cuda:0")
print("current", torch.cuda.current_stream().cuda_stream, "default", torch.cuda.default_stream().cuda_stream)
n = 1 << 27
x = torch.randn(n, device=dev); y = torch.empty_like(x)
torch.cuda.synchronize()
def lib_work(reps=30):
    for i in range(reps):
        ops.dropout(x, y, n, 0.5, seed=i)
for name, mk in (("torch.cuda.Event on current stream", lambda: torch.cuda.current_stream()),
                 ("ExternalStream(0)", lambda: torch.cuda.ExternalStream(0, device=dev)),
                 ("default_stream()", lambda: torch.cuda.default_stream())):
    lib_work()
    s = mk()
    ev = torch.cuda.Event(); ev.record(s)
    q0 = ev.query()
    t0 = time.perf_counter(); s.synchronize(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: event done right after record: {q0}; stream.synchronize took {1e3*(t1-t0):.2f} ms, then device sync {1e3*(t2-t1):.2f} ms")
# a side (non-blocking) torch stream waiting on an event recorded on the current stream after library work
lib_work()
side = torch.cuda.Stream()
ev = torch.cuda.Event(); ev.record()
side.wait_event(ev)
with torch.cuda.stream(side):
    z = y.clone()
t0 = time.perf_counter(); side.synchronize(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"side stream waited {1e3*(t1-t0):.2f} ms; remaining device work {1e3*(t2-t1):.2f} ms")
