cd $GRAFT_REPO_ROOT
for r in 1 2 3 4 5; do timeout 200 python attic/bimodal_probe2.py 2>&1 | tail -1; done
