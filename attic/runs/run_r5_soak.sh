set -x
cd $GRAFT_REPO_ROOT
B2T_AMP=1 timeout 300 python tools/stress_step.py 3000 2>&1 | tail -1
B2T_AMP=1 B2T_STRESS_SHAPE=c3 timeout 300 python tools/stress_step.py 3000 2>&1 | tail -1
timeout 300 python tools/stress_step.py 1500 2>&1 | tail -1
B2T_BWD_PAIRED=1 timeout 300 python tools/stress_step.py 1000 2>&1 | tail -1
B2T_STRESS_SHAPE=c3 timeout 300 python tools/stress_step.py 600 2>&1 | tail -1
