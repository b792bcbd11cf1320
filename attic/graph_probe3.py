import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nejm-brain-to-text_amd"))
import numpy as np, torch
import b2t_ops as ops
from rnn_model import GRUDecoder
dev = torch.device("cuda:0")
U, F, H, L, C, PATCH, STRIDE = 32, 512, 768, 5, 41, 14, 4
model = GRUDecoder(F, H, 4, C, 0.0, 0.0, L, PATCH, STRIDE).to(dev).eval()
day = torch.zeros(U, dtype=torch.int32, device=dev)
x = torch.randn(U, PATCH, F, device=dev) * 0.5
states = None
with torch.no_grad():
    for f in range(8):
        lg, states = model(x, day, states, True)
torch.cuda.synchronize()
ent = [e for e in model._graphs.values() if e["graph"] is not None][0]
g = ent["graph"]
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    host, tot = [], []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    return f"host p50 {np.percentile(host, 50) * 1e6:.1f} us, call+sync p50 {np.percentile(tot, 50) * 1e6:.1f} us"
print("replay only:", t(lambda: g.replay()))
print("query:", t(lambda: ent["done"].query()))
print("2x empty_like:", t(lambda: (torch.empty_like(ent["logits"]), torch.empty_like(ent["hidden"]))))
def tab():
    ti, to = ent["tab_in"], ent["tab_out"]
    ti[1], ti[2] = x.data_ptr(), day.data_ptr(); ti[3] = states.data_ptr(); to[5], to[6] = lg.data_ptr(), states.data_ptr()
print("table writes:", t(tab))
print("record:", t(lambda: ent["done"].record()))
print("replay + record:", t(lambda: (g.replay(), ent["done"].record())))
with torch.no_grad():
    print("model():", t(lambda: model(x, day, states, True)))
