# Which process-level quantity goes with the slow mode of a bimodal box?  n runs of the headline step: ms per step, the clocks
# and board power bench.py sampled in its timed region, the per-kernel breakdown, and the executor's queue calibration.
cd $GRAFT_REPO_ROOT
N=${1:-6}
for i in $(seq $N); do
  B2T_PLAN_DUMP=1 timeout 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2> gpurun_out/bim_$i.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; b=d['box']
print(d['ms_per_step'], 'sclk', b.get('sclk_mhz_p50'), 'W', b.get('board_w_p50'), 'enq', d['host_enqueue_ms_per_step'], {k: round(v,2) for k,v in r['breakdown_ms'].items() if v>1}, 'bwd_us', r['avg_launch_us'])"
  grep "exec: hop" gpurun_out/bim_$i.err | head -2
done
