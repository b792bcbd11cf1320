cd $GRAFT_REPO_ROOT
for rep in 1 2; do for w in f fb; do echo "WIDE=$w rep $rep"; B2T_WIDE_F32=$w timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['breakdown_ms']['gru_sweep_bwd'], d['roofline']['breakdown_ms']['gru_sweep_fwd'])"; done; done
export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out; cd /tmp
B2T_WIDE_F32=fb timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fb -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_pmc.py $OUT/pmc_fb/f_results.db | head -3; rm -rf $OUT/pmc_fb
