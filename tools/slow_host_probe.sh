# Does a slow HOST slow the step?  The headline bench alone, then sharing ONE core with a busy loop (taskset), then with the busy
# loop on the SMT sibling of its core.
cd $GRAFT_REPO_ROOT
run() { timeout 200 "$@" python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'enq', d['host_enqueue_ms_per_step'], 'W', d['box'].get('board_w_p50'))"; }
echo "alone:            $(run env X=0)"
echo "pinned to cpu 3:  $(run taskset -c 3)"
taskset -c 3 python -c "while True: pass" & HOG=$!
sleep 0.5
echo "cpu 3 + hog on 3: $(run taskset -c 3)"
kill $HOG
SIB=$(cat /sys/devices/system/cpu/cpu3/topology/thread_siblings_list | tr ',' '\n' | grep -v '^3$' | head -1)
taskset -c $SIB python -c "while True: pass" & HOG=$!
sleep 0.5
echo "cpu 3 + hog on sibling $SIB: $(run taskset -c 3)"
kill $HOG
echo "alone again:      $(run env X=0)"
