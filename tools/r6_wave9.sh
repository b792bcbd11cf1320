#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_wave10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -s > $O/pytest_wave.txt 2>&1; echo "pytest wave rc $?" | tee $O/summary.txt
tail -5 $O/pytest_wave.txt
export R6_PROBE_ONLY_TIMING=1
B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 600 python tools/r6_wave_probe.py > $O/probe_timing.txt 2>&1; grep "R6WAVE" $O/probe_timing.txt | tee -a $O/summary.txt; tail -2 $O/probe_timing.txt
timeout 600 python tools/r6_wave_probe.py > $O/probe.txt 2>&1; grep "R6WAVE" $O/probe.txt | tee -a $O/summary.txt
unset R6_PROBE_ONLY_TIMING
timeout 1500 python tools/r6_amp_ab.py 2>$O/amp_ab.err | grep R6AMP | tee -a $O/summary.txt; tail -3 $O/amp_ab.err
