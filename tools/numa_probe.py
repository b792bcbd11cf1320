#!/usr/bin/env python3
"""Is the box's per-process bimodality (19.3 vs 21 ms per step, host enqueue 2 vs 5-7 ms) NUMA placement?  Prints the GPU's
NUMA node and the nodes' CPU lists, then runs the headline bench pinned to each node's CPUs (and unpinned) a few times."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(os.path.join(d, "cpulist")).read())
print("nodes:", {n: f"{c[0]}..{c[-1]} ({len(c)} cpus)" for n, c in nodes.items()})
print("allowed cpus:", len(os.sched_getaffinity(0)), "current cpu:", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
gpus = []
for d in glob.glob("/sys/class/drm/card*/device"):
    try:
        vendor = open(os.path.join(d, "vendor")).read().strip()
        if vendor == "0x1002":
            gpus.append((os.path.realpath(d).rsplit("/", 1)[1], open(os.path.join(d, "numa_node")).read().strip()))
    except OSError:
        pass
print("amd gpus (pci, numa_node):", gpus)
try:
    print(subprocess.run(["cat", "/proc/self/status"], capture_output=True, text=True).stdout.split("Cpus_allowed_list")[1].splitlines()[0])
except Exception:      # noqa: BLE001
    pass


def run(cpus):
    code = ("import os,sys\n" + (f"os.sched_setaffinity(0, {sorted(cpus)!r})\n" if cpus else "") +
            f"sys.argv=['bench.py','--steps','30','--warmup','6','--no-cpu-baseline','--no-secondary']\n"
            f"sys.path.insert(0,{ROOT!r})\nimport runpy\nrunpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
        d = json.loads(out.strip().splitlines()[-1])
        return d["ms_per_step"], d["host_enqueue_ms_per_step"]
    except Exception as ex:      # noqa: BLE001
        return ("fail", str(ex)[:80])


for rep in range(3):
    print("unpinned:", run(None), flush=True)
    for n, c in nodes.items():
        allowed = sorted(set(c) & os.sched_getaffinity(0))
        if allowed:
            print(f"node {n} ({len(allowed)} cpus):", run(allowed), flush=True)
