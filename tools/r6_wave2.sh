#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_wave2; mkdir -p $O
timeout 600 python tools/r6_wave_probe.py > $O/probe_plain.txt 2>&1; grep "R6WAVE\|R6STALE" $O/probe_plain.txt | tee $O/summary.txt; tail -3 $O/probe_plain.txt
timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -s > $O/pytest_wave.txt 2>&1; echo "pytest wave rc $?" | tee -a $O/summary.txt
tail -15 $O/pytest_wave.txt
for d in 0 5 20; do B2T_EXEC_HOST_DELAY_US=$d timeout 300 python tools/r6_hostdelay_steps.py 2>&1 | grep R6DELAY | tee -a $O/summary.txt; done
B2T_EXEC_HOST_DELAY_US=5 R6_STEPS=200 timeout 300 python tools/r6_hostdelay_steps.py 2>&1 | grep R6DELAY | tee -a $O/summary.txt
python tools/r6_amp_ab.py 2>/dev/null | grep R6AMP | tee -a $O/summary.txt
