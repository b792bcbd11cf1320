#!/bin/bash
# Round 6, first call: (1) the GPU suite with per-test durations, (2) the SMI-poller A/B of the verdict's item 1c, (3) emulated host delay (item 1d).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r6_first
mkdir -p $O
export B2T_BENCH_NO_RESTART=1
B="python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 5"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "$tag $(python - <<PY
import json
try:
    d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1])
    print(d['ms_per_step'], d['config']['host_enqueue_ms_per_step'], d['config'].get('slow_mode_probe') and {k:d['config']['slow_mode_probe'].get(k) for k in ('hop_us_gpu','hop_us_host')})
except Exception as e: print('ERR', e)
PY
)" | tee -a $O/summary.txt; }
run plain1 X=1
run plain2 X=1 B2T_BENCH_FORCE_SLOW_PROBE=1
# SMI pollers (the driver's bench lease polls SMI every ~5 s)
( while true; do rocm-smi --json > /dev/null 2>&1; sleep 5; done ) & P1=$!
run rocmsmi_5s X=1
kill $P1
( while true; do rocm-smi --json > /dev/null 2>&1; sleep 0.5; done ) & P1=$!
run rocmsmi_0p5s X=1 B2T_BENCH_FORCE_SLOW_PROBE=1
kill $P1
( while true; do amd-smi metric --json > /dev/null 2>&1; sleep 5; done ) & P1=$!
run amdsmi_5s X=1
kill $P1
( while true; do amd-smi metric --json > /dev/null 2>&1; sleep 0.5; done ) & P1=$!
run amdsmi_0p5s X=1 B2T_BENCH_FORCE_SLOW_PROBE=1
kill $P1
( while true; do amd-smi metric --json > /dev/null 2>&1; rocm-smi --showuse --showmemuse --showpower --json > /dev/null 2>&1; done ) & P1=$!
run smi_busy X=1 B2T_BENCH_FORCE_SLOW_PROBE=1
kill $P1
# emulated slow host
for us in 5 10 20 40; do run hostdelay_$us B2T_EXEC_HOST_DELAY_US=$us B2T_BENCH_FORCE_SLOW_PROBE=1; done
run plain3 X=1
# (1) the suite
timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
tail -60 $O/pytest_gpu.txt
cat $O/summary.txt
