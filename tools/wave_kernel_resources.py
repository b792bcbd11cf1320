#!/usr/bin/env python3
"""Register / scratch use of every kernel of csrc/gru_wave.hip (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
The layer-wavefront kernels must not spill: partial reloads of spilled MFMA operand tuples were wrong on ROCm 7.2 (NOTES.md R6.2)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nejm-brain-to-text_amd", "csrc")


def resources(extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-x", "hip", "--cuda-device-only",
           "-c", os.path.join(CSRC, "gru_wave.hip"), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage", *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    res, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            try:
                cur = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip() or cur
            except OSError:
                pass
            cur = re.sub(r"^void b2t::", "", cur).split("(")[0]
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur:
            res[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    return res


if __name__ == "__main__":
    r = resources(sys.argv[1:])
    for k, v in r.items():
        print(f"{k:60s} VGPR {v.get('VGPRs', -1):3d}  AGPR {v.get('AGPRs', -1):3d}  scratch {v.get('ScratchSize', -1):4d}")
