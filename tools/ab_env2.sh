# like ab_env.sh, printing step ms, dominant-kernel launch us and roofline frac
cd $GRAFT_REPO_ROOT
REPS=$1; shift
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_us'], r['frac'], r['breakdown_ms'].get('gru_sweep_fwd'))"; }
for r in $(seq $REPS); do
echo "default: $(run A=1)   with $*: $(run "$@")"
done
