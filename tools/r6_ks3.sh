#!/bin/bash
# bf16-mode step A/B (K-split wavefront, both passes / forward only) + timeline of the default
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_ks3; mkdir -p $O
cat > /tmp/ab.py <<'PY'
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd"); sys.path.insert(0, ".")
import bench_secondary as bs
for shape, ks, dirs in (("c2", "1", "auto"), ("c3", "1", "auto"), ("c2", "1", "auto"), ("c3", "1", "auto")):
    os.environ["B2T_WAVE_KS"] = ks; os.environ["B2T_WAVE_DIRS"] = dirs
    try:
        r = bs.train_ms(shape, True)
        print("R6AMP", shape, "ks=" + ks, "dirs=" + dirs, r["ms_per_step"], r["window_ms"], flush=True)
    except Exception as e:
        print("R6AMP", shape, "ks=" + ks, "dirs=" + dirs, "ERROR", repr(e)[:300], flush=True)
PY
timeout 1200 python /tmp/ab.py 2>$O/ab.err | grep R6AMP | tee $O/summary.txt; tail -3 $O/ab.err
