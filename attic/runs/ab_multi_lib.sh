# A/B several builds of the library inside one gpurun call: ab_multi_lib.sh reps lib1.so lib2.so ...   (paths relative to csrc/)
cd $GRAFT_REPO_ROOT
REPS=$1; shift
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
for i in $(seq $REPS); do
  for L in "$@"; do echo "$L: $(run B2T_LIB=$GRAFT_REPO_ROOT/nejm-brain-to-text_amd/csrc/$L)"; done
done
