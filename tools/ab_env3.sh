# A/B environment settings inside one gpurun call: ab_env3.sh reps "VAR=a" "VAR=b" ...   (use X=0 for the default)
cd $GRAFT_REPO_ROOT
REPS=$1; shift
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'], d['roofline']['breakdown_ms'].get('gru_sweep_fwd'))"; }
for i in $(seq $REPS); do
  for E in "$@"; do echo "$E: $(run $E)"; done
done
