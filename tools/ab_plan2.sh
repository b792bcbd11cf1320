cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
echo "default (3 workers, sweeps on workers, 6/4):  $(run A=1)"
echo "wgrad per chunk all layers:                    $(run B2T_WGRAD_CHUNK_MASK=31)"
echo "sweeps on any queue:                           $(run B2T_SWEEP_ANYQ=1)"
echo "sweeps any + wgrad per chunk:                  $(run B2T_SWEEP_ANYQ=1 B2T_WGRAD_CHUNK_MASK=31)"
echo "8/6:                                           $(run B2T_CHUNKS=8 B2T_CHUNKS_BWD=6)"
echo "8/6 wgrad per chunk:                           $(run B2T_CHUNKS=8 B2T_CHUNKS_BWD=6 B2T_WGRAD_CHUNK_MASK=31)"
echo "10/8 wgrad per chunk:                          $(run B2T_CHUNKS=10 B2T_CHUNKS_BWD=8 B2T_WGRAD_CHUNK_MASK=31)"
echo "10/8 any, per chunk:                           $(run B2T_CHUNKS=10 B2T_CHUNKS_BWD=8 B2T_WGRAD_CHUNK_MASK=31 B2T_SWEEP_ANYQ=1)"
echo "4 workers:                                     $(run B2T_WORKERS=4)"
echo "2 workers any:                                 $(run B2T_WORKERS=2 B2T_SWEEP_ANYQ=1)"
