#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_rgf3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -s > $O/pytest_wave.txt 2>&1; echo "pytest wave rc $?" | tee $O/summary.txt
tail -5 $O/pytest_wave.txt
B2T_WAVE_RGF=0 timeout 900 python -m pytest tests/test_gpu_wave.py -x -q > $O/pytest_wave_fat.txt 2>&1; echo "pytest wave (16-unit form) rc $?" | tee -a $O/summary.txt
export R6_PROBE_ONLY_TIMING=1
B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 600 python tools/r6_wave_probe.py > $O/probe_timing.txt 2>&1; grep "R6WAVE" $O/probe_timing.txt | tee -a $O/summary.txt; tail -2 $O/probe_timing.txt
B2T_WAVE_RGF=0 B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 600 python tools/r6_wave_probe.py > $O/probe_timing_fat.txt 2>&1; grep "R6WAVE" $O/probe_timing_fat.txt | sed 's/R6WAVE/R6WAVE_FAT/' | tee -a $O/summary.txt
unset R6_PROBE_ONLY_TIMING
cat > /tmp/ab.py <<'PY'
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd"); sys.path.insert(0, ".")
import bench_secondary as bs
for shape, rgf, dirs, chunks in (("c2", "1", "f", "1,1"), ("c2", "0", "f", "1,1"), ("c2", "1", "fb", "1,1"), ("c2", "0", "fb", "1,1"), ("c3", "0", "f", "1,1"), ("c3", "0", "fb", "1,1"), ("c2", "1", "f", "1,1"), ("c2", "0", "f", "1,1")):
    os.environ["B2T_WAVE_RGF"] = rgf; os.environ["B2T_WAVE_DIRS"] = dirs; os.environ["B2T_WAVE_CHUNKS"] = chunks
    r = bs.train_ms(shape, True)
    print("R6AMP", shape, "rgf=" + rgf, "dirs=" + dirs, "chunks=" + chunks, r["ms_per_step"], r["window_ms"], flush=True)
PY
timeout 1200 python /tmp/ab.py 2>$O/ab.err | grep R6AMP | tee -a $O/summary.txt; tail -3 $O/ab.err
