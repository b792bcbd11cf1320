set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/r5_suite.log 2>&1; tail -15 $OUT/r5_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
