set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_step_parity.py -x -q -m gpu -k "paired" 2>&1 | tail -15
for v in 0 1 0 1; do
B2T_BWD_PAIRED=$v B2T_BENCH_NO_RESTART=1 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('PAIRED=$v', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['roofline']['breakdown_ms'])"
done
