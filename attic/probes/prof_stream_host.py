import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import torch
from rnn_model import GRUDecoder
dev = torch.device("cuda:0")
model = GRUDecoder(512, 768, 4, 41, 0.0, 0.0, 5, 14, 4).to(dev).eval()
day = torch.zeros(32, dtype=torch.int32, device=dev)
x = torch.randn(32, 14, 512, device=dev)
states = None
with torch.no_grad():
    for _ in range(20):
        lg, states = model(x, day, states, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        lg, states = model(x, day, states, True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue per step {1e3*(t1-t0)/100:.3f} ms, incl. drain {1e3*(t2-t0)/100:.3f} ms")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100):
        lg, states = model(x, day, states, True)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
