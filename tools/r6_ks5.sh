#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_ks5; mkdir -p $O; : > $O/summary.txt
export R6_PROBE_ONLY_TIMING=1 R6_PROBE_CFGS="5,500,64,512,0.0;1,500,64,512,0.0"
for v in wtiming; do
  B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_$v.so timeout 600 python tools/r6_wave_probe.py 2>&1 | grep R6WAVE | sed "s/R6WAVE/R6WAVE $v/" | cut -c1-1500 | tee -a $O/summary.txt
done
