cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
export B2T_SWEEP_ANYQ=1
echo "any 6/4:          $(run A=1)"
echo "any 6/4 narrow f: $(run B2T_WIDE_F32=)"
echo "any 6/4 wide fb:  $(run B2T_WIDE_F32=fb)"
echo "any 5/4:          $(run B2T_CHUNKS=5)"
echo "any 7/4:          $(run B2T_CHUNKS=7)"
echo "any 8/4:          $(run B2T_CHUNKS=8)"
echo "any 6/3:          $(run B2T_CHUNKS_BWD=3)"
echo "any 6/5:          $(run B2T_CHUNKS_BWD=5)"
echo "any 6/6:          $(run B2T_CHUNKS_BWD=6)"
echo "any 6/4 wg0:      $(run B2T_WGRAD_CHUNK_MASK=1)"
echo "any 6/4 wg01:     $(run B2T_WGRAD_CHUNK_MASK=3)"
echo "any 6/4:          $(run A=1)"
