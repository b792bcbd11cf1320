# A/B two builds of the library inside one gpurun call: ab_lib.sh <old.so> [reps]
cd $GRAFT_REPO_ROOT
OLD=$1; REPS=${2:-3}
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in $(seq $REPS); do
echo "new: $(run A=1)   old: $(run B2T_LIB=$GRAFT_REPO_ROOT/$OLD)"
done
