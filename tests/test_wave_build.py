"""The layer-wavefront kernels (csrc/gru_wave.hip, gru_wave_ks.h) must compile WITHOUT scratch: with spills, ROCm 7.2's allocator
split 16-byte MFMA operand tuples (weight fragments pinned to accumulation registers, fragments in flight) into partial spills whose
reloads came back wrong -- deterministic NaNs in the backward wavefront (NOTES.md R6.2).  A spill is therefore a build error here."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_wavefront_kernels_do_not_spill():
    import wave_kernel_resources as W
    res = W.resources()
    ks = [k for k in res if "gru_ks_" in k]
    assert len(ks) == 16 and any("gru_wave_fwd_kernel" in k for k in res) and any("gru_wave_bwd_kernel" in k for k in res), sorted(res)
    spilled = {k: v for k, v in res.items() if v.get("ScratchSize", -1) != 0}
    assert not spilled, spilled
    # one workgroup per CU is what the placement assumes for the large shapes: their registers must allow at least one wave per SIMD
    assert all(v.get("VGPRs", 0) <= 256 and v.get("AGPRs", 0) <= 256 for v in res.values())
