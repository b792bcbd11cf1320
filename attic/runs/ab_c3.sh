cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 120 python tools/bench_c3.py 2>&1 | tail -1 | cut -c52-90; }
for c in "2 2" "3 2" "3 3" "4 2" "4 3" "5 3" "6 3" "6 4" "3 2"; do set -- $c; echo "amp wide $1/$2: $(run B2T_AMP=1 B2T_CHUNKS=$1 B2T_CHUNKS_BWD=$2)"; done
