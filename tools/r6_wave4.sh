#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_wave4; mkdir -p $O
export R6_PROBE_ONLY_TIMING=1
B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 600 python tools/r6_wave_probe.py > $O/probe_timing.txt 2>&1; grep "R6WAVE" $O/probe_timing.txt | tee $O/summary.txt; tail -2 $O/probe_timing.txt
B2T_WAVE_LOCAL=0 B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 600 python tools/r6_wave_probe.py > $O/probe_timing_nolocal.txt 2>&1; grep "R6WAVE" $O/probe_timing_nolocal.txt | tee -a $O/summary.txt
# verdict 1d: what in bench.py makes the emulated slow host cost 1 ms where the bare loop (tools/r6_hostdelay_steps.py) loses 0.1
export B2T_BENCH_NO_RESTART=1
B="python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 5"
run() { tag=$1; shift; env "$@" $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][-1]); print('BENCH $tag', d['ms_per_step'], d['config']['host_enqueue_ms_per_step'])
except Exception as e: print('BENCH $tag ERR', e)
PY
}
run plain X=1
run delay5 B2T_EXEC_HOST_DELAY_US=5
run delay5_nosampler B2T_EXEC_HOST_DELAY_US=5 B2T_BENCH_NO_SAMPLER=1
run delay5_noprobe B2T_EXEC_HOST_DELAY_US=5 B2T_BENCH_NO_PROBE=1
run delay5_neither B2T_EXEC_HOST_DELAY_US=5 B2T_BENCH_NO_PROBE=1 B2T_BENCH_NO_SAMPLER=1
run plain_neither B2T_BENCH_NO_PROBE=1 B2T_BENCH_NO_SAMPLER=1
B="python bench.py --no-secondary --no-cpu-baseline --steps 120 --warmup 5"; run delay5_steps120 B2T_EXEC_HOST_DELAY_US=5
