cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "6/3:   $(run B2T_CHUNKS_BWD=3)"
echo "8/3:   $(run B2T_CHUNKS=8 B2T_CHUNKS_BWD=3)"
echo "7/3:   $(run B2T_CHUNKS=7 B2T_CHUNKS_BWD=3)"
echo "10/3:  $(run B2T_CHUNKS=10 B2T_CHUNKS_BWD=3)"
echo "6/2:   $(run B2T_CHUNKS_BWD=2)"
echo "8/2:   $(run B2T_CHUNKS=8 B2T_CHUNKS_BWD=2)"
echo "6/3:   $(run B2T_CHUNKS_BWD=3)"
echo "8/3:   $(run B2T_CHUNKS=8 B2T_CHUNKS_BWD=3)"
