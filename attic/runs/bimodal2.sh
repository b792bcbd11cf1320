# which processes are slow?  N short bench runs with the queue calibration printed (B2T_PLAN_DUMP) and the per-kernel breakdown
cd $GRAFT_REPO_ROOT
N=${1:-6}
for r in $(seq $N); do
  B2T_PLAN_DUMP=1 timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2> /tmp/err_$r.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms', d['ms_per_step'], 'enq', d['host_enqueue_ms_per_step'], 'box', d.get('box'), 'bd', {k[:14]:v for k,v in list(r['breakdown_ms'].items())[:5]})"
  grep -i "calib\|worker\|hop\|queue" /tmp/err_$r.txt | grep -v "^  " | head -8
done
