cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; rm -rf $OUT/tl_c3amp
(cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace -d $OUT/tl_c3amp -o tl -- python tools/r4_cfgs.py c3_amp > $OUT/tl_c3amp.log 2>&1)
cd $GRAFT_REPO_ROOT
python tools/rocprof_timeline.py $OUT/tl_c3amp/tl_results.db 3 1 full > $OUT/tl_c3amp_full3.txt
head -16 $OUT/tl_c3amp_full3.txt
rm -rf $OUT/tl_c3amp
