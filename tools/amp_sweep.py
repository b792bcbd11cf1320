#!/usr/bin/env python3
"""One gpurun call: the bf16-operand step (C2 and the shipped C3 shape) under a grid of plan knobs, each in its own process."""
import itertools, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys,json; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench_secondary as S; "
        "r = S.train_ms(sys.argv[1], True, steps=16, warmup=4); print(json.dumps(r['ms_per_step']))") % (os.path.join(ROOT, "tools"), ROOT)


def run(shape, env):
    e = dict(os.environ); e.update({k: str(v) for k, v in env.items()})
    try:
        out = subprocess.run([sys.executable, "-c", CODE, shape], env=e, capture_output=True, text=True, timeout=300)
        return float(out.stdout.strip().splitlines()[-1])
    except Exception as ex:      # noqa: BLE001
        return 1e9


for shape in ("c3", "c2"):
    print(shape, "default:", [run(shape, {}) for _ in range(2)], flush=True)
    res = []
    for cf, cb in itertools.product((1, 2, 3, 4, 6), (1, 2, 3, 4)):
        ms = run(shape, {"B2T_CHUNKS": cf, "B2T_CHUNKS_BWD": cb})
        res.append((ms, cf, cb)); print(f"  {shape} chunks {cf}/{cb}: {ms:.3f}", flush=True)
    for k, v in (("B2T_AMP_WIDE", "f"), ("B2T_AMP_WIDE", "b"), ("B2T_AMP_WIDE", ""), ("B2T_GRU_LOCAL", ""), ("B2T_SPLITK_TARGET", 384), ("B2T_SPLITK_TARGET", 768)):
        print(f"  {shape} {k}={v!r}: {run(shape, {k: v}):.3f}", flush=True)
    res.sort()
    print(shape, "best three again:", [(cf, cb, [round(run(shape, {'B2T_CHUNKS': cf, 'B2T_CHUNKS_BWD': cb}), 3) for _ in range(2)]) for _, cf, cb in res[:3]], flush=True)
