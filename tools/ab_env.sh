# usage: ab_env.sh REPS ENV=VAL [ENV=VAL ...] : headline step time with and without the given environment, alternating processes
cd $GRAFT_REPO_ROOT
REPS=$1; shift
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_us'], d['final_loss'])"; }
for r in $(seq $REPS); do
echo "default: $(run A=1)   with $*: $(run "$@")"
done
