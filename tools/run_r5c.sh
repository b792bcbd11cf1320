set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "paired" 2>&1 | tail -15
timeout 600 python tools/r5_pair_probe.py > $OUT/r5c_probe.log 2>&1; cat $OUT/r5c_probe.log
