#!/bin/bash
# Round 6: the layer wavefront on the box -- kernel tests, launch times alone, then the bf16 steps with and without it.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_wave; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -s > $O/pytest_wave.txt 2>&1; echo "pytest wave rc $?" | tee $O/summary.txt
tail -25 $O/pytest_wave.txt
timeout 600 python tools/r6_wave_probe.py > $O/probe.txt 2>&1; grep R6WAVE $O/probe.txt | tee -a $O/summary.txt; tail -3 $O/probe.txt
cat > /tmp/amp_ab.py <<'PY'
import json, os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd"); sys.path.insert(0, ".")
import bench_secondary as bs
for shape in ("c3", "c2"):
    for wave in ("1", "0", "1"):
        os.environ["B2T_WAVE"] = wave
        try:
            r = bs.train_ms(shape, True)
            print("R6AMP", shape, "wave=" + wave, r["ms_per_step"], r["window_ms"], flush=True)
        except Exception as e:
            print("R6AMP", shape, "wave=" + wave, "ERROR", repr(e)[:300], flush=True)
PY
timeout 900 python /tmp/amp_ab.py > $O/amp_ab.txt 2>&1; grep R6AMP $O/amp_ab.txt | tee -a $O/summary.txt; tail -5 $O/amp_ab.txt
# timelines of the fp32 headline step, healthy against an emulated slow host (verdict item 1d)
export B2T_BENCH_NO_RESTART=1
bash tools/prof_tl.sh r6_fp32_plain X=1 > /dev/null 2>&1
bash tools/prof_tl.sh r6_fp32_delay5 B2T_EXEC_HOST_DELAY_US=5 > /dev/null 2>&1
head -3 gpurun_out/tl_r6_fp32_plain.txt gpurun_out/tl_r6_fp32_delay5.txt
grep -h "ms_per_step" gpurun_out/tl_r6_fp32_plain.log gpurun_out/tl_r6_fp32_delay5.log | cut -c1-200
