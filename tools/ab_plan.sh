cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
for fb in "6 4" "6 3" "6 2" "7 4" "8 4" "8 3" "5 4" "5 3" "7 3"; do set -- $fb; echo "fwd$1 bwd$2: $(run B2T_CHUNKS=$1 B2T_CHUNKS_BWD=$2)"; done
done
