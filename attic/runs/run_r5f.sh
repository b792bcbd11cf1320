# Round-5 GPU call f: where does the shipped shape's bf16 step (c3_amp) spend its 6 ms?  Full kernel timeline of one step + variants.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp
rm -rf $OUT/tl_c3amp
(cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace -d $OUT/tl_c3amp -o tl -- python tools/r4_cfgs.py c3_amp > $OUT/tl_c3amp.log 2>&1)
cd $GRAFT_REPO_ROOT
python tools/rocprof_timeline.py $OUT/tl_c3amp/tl_results.db 3 1 full > $OUT/tl_c3amp_full.txt
head -20 $OUT/tl_c3amp_full.txt
rm -rf $OUT/tl_c3amp
rm -rf $OUT/tl_c2amp
cd /tmp
(cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace -d $OUT/tl_c2amp -o tl -- python tools/r4_cfgs.py c2_amp > $OUT/tl_c2amp.log 2>&1)
cd $GRAFT_REPO_ROOT
python tools/rocprof_timeline.py $OUT/tl_c2amp/tl_results.db 3 1 full > $OUT/tl_c2amp_full.txt
head -20 $OUT/tl_c2amp_full.txt
rm -rf $OUT/tl_c2amp
for e in "X=1" "B2T_GRU_LOCAL_MAXH=768" "B2T_CHUNKS=4 B2T_CHUNKS_BWD=3" "B2T_CHUNKS=2 B2T_CHUNKS_BWD=2" "B2T_WGRAD_CHUNK_MASK=31"; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1
done
