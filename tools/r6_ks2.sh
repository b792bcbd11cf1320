#!/bin/bash
# K-split form: per-phase cycles only (timing build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_ks2; mkdir -p $O
export R6_PROBE_ONLY_TIMING=1 R6_PROBE_CFGS="${R6_PROBE_CFGS:-5,500,64,512,0.0;1,500,64,512,0.0}"
B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 900 python tools/r6_wave_probe.py > $O/probe.txt 2>&1; grep "R6WAVE" $O/probe.txt | cut -c1-1800 | tee $O/summary.txt; tail -3 $O/probe.txt | grep -v R6WAVE
