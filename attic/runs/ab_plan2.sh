cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_us'])"; }
for r in 1 2; do
echo "default:        $(run A=1)"
echo "wide bwd too:   $(run B2T_WIDE_F32=fb)"
echo "narrow fwd:     $(run B2T_WIDE_F32=)"
echo "8/4:            $(run B2T_CHUNKS=8)"
echo "6/3:            $(run B2T_CHUNKS_BWD=3)"
echo "6/5:            $(run B2T_CHUNKS_BWD=5)"
echo "8/5:            $(run B2T_CHUNKS=8 B2T_CHUNKS_BWD=5)"
done
