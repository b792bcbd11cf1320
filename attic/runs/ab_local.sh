cd $GRAFT_REPO_ROOT
for m in 1 1025 3073; do echo mode $m; B2T_DIR=bwd B2T_NS=1,2 timeout 100 python tools/bench_sweep.py 250 $m 2>&1 | grep "parity\|error word\|bwd N"; done
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_us'], r['breakdown_ms'].get('gru_sweep_fwd'), r['breakdown_ms'].get('gru_sweep_bwd'))"; }
for r in 1 2 3; do
echo "default: $(run A=1)   local b: $(run B2T_GRU_LOCAL=b)   local fb: $(run B2T_GRU_LOCAL=fb)"
done
B2T_GRU_LOCAL=fb timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_step_parity.py -q -x 2>&1 | tail -2
