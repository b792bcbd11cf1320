# A/B environment settings inside one gpurun call: ab_env3.sh reps "VAR=a" "VAR=b" ...   (use X=0 for the default)
cd $GRAFT_REPO_ROOT
REPS=$1; shift
run() { env "$@" timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['box']; r=d['roofline']['breakdown_ms']; print(d['ms_per_step'], d['final_loss'], 'sclk', b.get('sclk_mhz_p50'), 'W', b.get('board_w_p50'), 'enq', d['host_enqueue_ms_per_step'], 'fwd', r.get('gru_sweep_fwd'), 'bwd', r.get('gru_sweep_bwd'), 'g00', r.get('gemm_f32_kernel<0,0>'), 'g10', r.get('gemm_f32_kernel<1,0>'), 'g11', r.get('gemm_f32_kernel<1,1>'), 'ctc', r.get('ctc_kernel'))"; }
for i in $(seq $REPS); do
  for E in "$@"; do echo "$E: $(run $E)"; done
done
