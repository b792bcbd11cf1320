cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['breakdown_ms'].get('gru_sweep_fwd'))"; }
for rep in 1 2; do
echo "base:        $(run A=1)"
echo "fwd sweeps 4: $(run B2T_FWD_SWEEP_STREAMS=4)"
echo "fwd sweeps 3: $(run B2T_FWD_SWEEP_STREAMS=3)"
echo "narrow fwd:   $(run B2T_WIDE_F32=)"
echo "narrow 4:     $(run B2T_WIDE_F32= B2T_FWD_SWEEP_STREAMS=4)"
done
