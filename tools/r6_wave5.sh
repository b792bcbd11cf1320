#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_wave5; mkdir -p $O
export R6_PROBE_ONLY_TIMING=1
B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 600 python tools/r6_wave_probe.py > $O/probe_timing.txt 2>&1; grep "R6WAVE" $O/probe_timing.txt | tee $O/summary.txt; tail -2 $O/probe_timing.txt
B2T_WAVE_LOCAL=0 B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 600 python tools/r6_wave_probe.py > $O/probe_timing_nolocal.txt 2>&1; grep "R6WAVE" $O/probe_timing_nolocal.txt | tee -a $O/summary.txt
