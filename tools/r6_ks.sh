#!/bin/bash
# K-split form of the wavefront (gru_wave_ks.h): kernel tests, stale-workspace check, per-phase cycles, bf16-mode step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_ks; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_wave.py -x -q > $O/pytest_wave.txt 2>&1; echo "pytest wave rc $?" | tee $O/summary.txt
tail -15 $O/pytest_wave.txt
B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 900 python tools/r6_wave_probe.py > $O/probe_ks.txt 2>&1; grep "R6WAVE\|R6STALE" $O/probe_ks.txt | cut -c1-1500 | tee -a $O/summary.txt; tail -3 $O/probe_ks.txt
cat > /tmp/ab.py <<'PY'
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd"); sys.path.insert(0, ".")
import bench_secondary as bs
os.environ["B2T_WAVE_RGF"] = "0"
for shape, ks, dirs in (("c2", "1", "f"), ("c2", "0", "f"), ("c2", "1", "fb"), ("c2", "0", "fb"), ("c2", "1", "f"), ("c2", "1", "fb")):
    os.environ["B2T_WAVE_KS"] = ks; os.environ["B2T_WAVE_DIRS"] = dirs
    try:
        r = bs.train_ms(shape, True)
        print("R6AMP", shape, "ks=" + ks, "dirs=" + dirs, r["ms_per_step"], r["window_ms"], flush=True)
    except Exception as e:
        print("R6AMP", shape, "ks=" + ks, "dirs=" + dirs, "ERROR", repr(e)[:300], flush=True)
PY
timeout 1200 python /tmp/ab.py 2>$O/ab.err | grep R6AMP | tee -a $O/summary.txt; tail -3 $O/ab.err
