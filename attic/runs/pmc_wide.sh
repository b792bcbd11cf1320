cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf $OUT/pmc_w_$c
B2T_WIDE_F32=fb timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_w_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_w_$c.log 2>&1
echo "== $c"; python $GRAFT_REPO_ROOT/tools/rocprof_pmc.py $OUT/pmc_w_$c/p_results.db | grep "gru_persist"
rm -rf $OUT/pmc_w_$c
done
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
for r in 1 2 3; do echo "narrow bwd: $(run A=1)    wide bwd: $(run B2T_WIDE_F32=fb)"; done
