"""Random shapes through every GRU sweep variant (tools/r4_fuzz_sweeps.py): the persistent kernels against the step-launch ones,
XCD-local against device scope, 32- against 16-unit workgroups, the fused forward against the plain one + a torch projection, the
bf16 fragment hand-off against fp32 tiles -- bit-identical wherever the kernels promise it.  (Round 4: a comparison of this kind
found the XCD-local hand-off returning stale tiles at H = 48, B = 5.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [5, 23])
def test_sweep_variants_agree_on_random_shapes(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r4_fuzz_sweeps.py"), "70", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "70 cases, 0 with differences" in r.stdout, r.stdout[-3000:]
