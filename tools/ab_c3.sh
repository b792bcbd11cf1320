cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 120 python tools/bench_c3.py 2>&1 | tail -1 | cut -c52-100; }
echo "amp serial:        $(run B2T_AMP=1)"
echo "amp 2/2:           $(run B2T_AMP=1 B2T_CHUNKS=2 B2T_CHUNKS_BWD=2)"
echo "amp 2/1:           $(run B2T_AMP=1 B2T_CHUNKS=2 B2T_CHUNKS_BWD=1)"
echo "amp 3/1:           $(run B2T_AMP=1 B2T_CHUNKS=3 B2T_CHUNKS_BWD=1)"
echo "amp 3/2:           $(run B2T_AMP=1 B2T_CHUNKS=3 B2T_CHUNKS_BWD=2)"
echo "amp 4/2:           $(run B2T_AMP=1 B2T_CHUNKS=4 B2T_CHUNKS_BWD=2)"
echo "amp serial:        $(run B2T_AMP=1)"
