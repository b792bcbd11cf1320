# usage: ab_multi.sh REPS "ENV=VAL ..." "ENV=VAL ..." ... : headline step time under each environment, interleaved
cd $GRAFT_REPO_ROOT
REPS=$1; shift
run() { env $1 timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for r in $(seq $REPS); do
  line=""
  for e in "$@"; do line="$line  [$e] $(run "$e")"; done
  echo "$line"
done
