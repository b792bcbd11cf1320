#!/bin/bash
# Round 6 (verdict 1d): which runtime CALLS take the host's time in a slow process?  bench processes with the executor's host timing on.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_ht; mkdir -p $O
export B2T_BENCH_NO_RESTART=1 B2T_EXEC_HOST_TIMING=1
B="python bench.py --no-secondary --no-cpu-baseline --steps 40 --warmup 6"
run() { tag=$1; shift; env "$@" $B > $O/b.json 2> $O/b.err; python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print('HT $tag', d['ms_per_step'], 'enq', d['config']['host_enqueue_ms_per_step'], 'dp_call_ms', d['config'].get('dp_collective_call_host_ms_per_step'))
except Exception as e: print('HT $tag ERR', e, open('$O/b.err').read()[-300:])
PY
grep "exec host timing" $O/b.err | tail -1 | cut -c1-600 | tee -a $O/summary.txt; }
for i in 1 2 3 4 5 6 7 8; do run plain$i X=1; done
run late3_a B2T_DP_FORCE=1 B2T_DP_TEST_DELAY_US=500
run late3_b B2T_DP_FORCE=1 B2T_DP_TEST_DELAY_US=500
run late3_est600 B2T_DP_FORCE=1 B2T_DP_TEST_DELAY_US=500 B2T_BUCKET_EST_US=600
run forced B2T_DP_FORCE=1
