#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_dp2; mkdir -p $O
export B2T_BENCH_NO_RESTART=1
B="python bench.py --no-secondary --no-cpu-baseline --steps 30 --warmup 6"
run() { tag=$1; shift; for rep in 1 2 3; do env "$@" $B > $O/b.json 2> $O/b.err; python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print('DP $tag', d['ms_per_step'], 'enq', d['config']['host_enqueue_ms_per_step'])
except Exception as e: print('DP $tag ERR', e, open('$O/b.err').read()[-300:])
PY
done; }
run plain X=1
run forced_est600 B2T_DP_FORCE=1 B2T_BUCKET_EST_US=600
run forced_est1500 B2T_DP_FORCE=1 B2T_BUCKET_EST_US=1500
run late_3_est1000 B2T_DP_FORCE=1 B2T_DP_TEST_DELAY_US=500 B2T_BUCKET_EST_US=1000
run late_3_est1500 B2T_DP_FORCE=1 B2T_DP_TEST_DELAY_US=500 B2T_BUCKET_EST_US=1500
run late_3_est3000 B2T_DP_FORCE=1 B2T_DP_TEST_DELAY_US=500 B2T_BUCKET_EST_US=3000
run late_8_est600 B2T_DP_FORCE=1 B2T_DP_TEST_DELAY_US=500 B2T_BUCKET_EST_US=600 B2T_DP_COALESCE=0
bash tools/run_prof_r6.sh > $O/prof.log 2>&1; tail -5 $O/prof.log
