cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['breakdown_ms'].get('gru_sweep_fwd'))"; }
for r in 1 2 3; do
echo "default: $(run A=1)     local fwd: $(run B2T_GRU_LOCAL=1)"
done
