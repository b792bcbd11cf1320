# A/B environment settings for the shipped-shape bf16 step inside one gpurun call: ab_c3amp.sh reps "VAR=a" ...
cd $GRAFT_REPO_ROOT
REPS=$1; shift
run() { env B2T_AMP=1 "$@" timeout 200 python tools/bench_c3.py 2>/dev/null | tail -1 | cut -c1-140; }
for i in $(seq $REPS); do
  for E in "$@"; do echo "$E: $(run $E)"; done
done
