#!/usr/bin/env python3
"""One gpurun call: the headline step under a grid of plan knobs (scheduler estimates, chunk counts, split-K target ...), one run
each, then the best few repeated.  Usage: plan_sweep.py [quick]"""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(env):
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "6", "--no-cpu-baseline", "--no-secondary"],
                             env=e, capture_output=True, text=True, timeout=200).stdout
        d = json.loads(out.strip().splitlines()[-1])
        return d["ms_per_step"], d["final_loss"]
    except Exception as ex:       # noqa: BLE001
        return 1e9, str(ex)[:60]


grid = []
for tfs, f, b in itertools.product((40, 80, 140), (4.0, 5.5, 7.5), (4.5, 6.0, 8.0)):
    grid.append({"B2T_EST_GEMM_TFS": tfs, "B2T_EST_FWD_US": f, "B2T_EST_BWD_US": b})
for sk in (512, 640, 896, 1024):
    grid.append({"B2T_SPLITK_TARGET": sk})
for ne, ch in ((2, 6), (2, 5), (1, 5), (3, 7)):
    grid.append({"B2T_NARROW_EDGE": ne, "B2T_CHUNKS": ch})
for cb in (3, 4, 5):
    grid.append({"B2T_CHUNKS_BWD": cb})
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    grid = grid[:3]
base = [run({}) for _ in range(3)]
print("default:", base, flush=True)
res = []
for g in grid:
    ms, loss = run(g)
    res.append((ms, g, loss))
    print(f"{ms:8.3f}  {g}  {loss}", flush=True)
res.sort(key=lambda r: r[0])
print("---- best five, three more runs each (and the default again) ----", flush=True)
for ms, g, _ in res[:5]:
    again = [run(g)[0] for _ in range(3)]
    print(f"{g}: first {ms:.3f}, then {again}", flush=True)
print("default again:", [run({})[0] for _ in range(3)], flush=True)
