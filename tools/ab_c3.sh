cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 120 python tools/bench_c3.py 2>&1 | tail -1 | cut -c1-110; }
echo "amp serial:        $(run B2T_AMP=1)"
echo "amp 2/2:           $(run B2T_AMP=1 B2T_CHUNKS=2 B2T_CHUNKS_BWD=2)"
echo "amp 3/2:           $(run B2T_AMP=1 B2T_CHUNKS=3 B2T_CHUNKS_BWD=2)"
echo "amp 4/3:           $(run B2T_AMP=1 B2T_CHUNKS=4 B2T_CHUNKS_BWD=3)"
echo "amp 6/4:           $(run B2T_AMP=1 B2T_CHUNKS=6 B2T_CHUNKS_BWD=4)"
echo "amp 4/3 workers-only: $(run B2T_AMP=1 B2T_CHUNKS=4 B2T_CHUNKS_BWD=3 B2T_SWEEP_WORKERS_ONLY=1)"
echo "f32 serial:        $(run A=1)"
echo "f32 2/2:           $(run B2T_CHUNKS=2 B2T_CHUNKS_BWD=2)"
echo "f32 3/2:           $(run B2T_CHUNKS=3 B2T_CHUNKS_BWD=2)"
