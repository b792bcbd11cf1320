# like ab_env.sh, printing step ms, host enqueue ms and the per-kernel busy times of the instrumented steps
cd $GRAFT_REPO_ROOT
REPS=$1; shift
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], 'enq', d['host_enqueue_ms_per_step'], 'launch_us', r['avg_launch_us'], ' '.join(f'{k[-6:]}={v}' for k,v in r['breakdown_ms'].items()))"; }
for r in $(seq $REPS); do
echo "default: $(run A=1)"
echo "with $*: $(run "$@")"
done
