cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_us'])"; }
for r in 1 2 3 4; do
echo "device-scope: $(run A=1)   xcd-local: $(run B2T_GRU_LOCAL=fb)"
done
B2T_GRU_LOCAL=fb timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_step_parity.py tests/test_gpu_parity.py -q -x 2>&1 | tail -2
