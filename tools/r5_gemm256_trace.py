#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv of tools/r5_gemm256.py: per (shape, kernel) the GEMM kernel's mean duration over the 10 timed calls."""
import csv, glob, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
SH = (("dW_ih0 2304x7168x7808", 2304, 7168, 7808), ("dX0 3904x7168x2304", 3904, 7168, 2304), ("dX0 whole 7808x7168x2304", 7808, 7168, 2304),
      ("gi0 whole 7808x2304x7168", 7808, 2304, 7168), ("gi0 chunk 2624x2304x7168", 2624, 2304, 7168), ("4096^3", 4096, 4096, 4096),
      ("odd 1000x3000x520", 1000, 3000, 520), ("gi l>=1 chunk 2624x2304x768", 2624, 2304, 768), ("dX l>=1 chunk 3904x768x2304", 3904, 768, 2304),
      ("c2 gi chunk 5376x1536x512", 5376, 1536, 512), ("c2 dX chunk 8000x512x1536", 8000, 512, 1536), ("c2 dW_ih 1536x512x32000", 1536, 512, 32000))
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_bf16p_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i = 0
for name, M, N, K in SH:
    out = []
    for mode in range(2):
        chunk = rows[i:i + 13]; i += 13
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in chunk[3:]]
        us = sum(d) / len(d) / 1e3
        nm = chunk[0]["Kernel_Name"]
        kn = "256" if "256" in nm else "128"
        out.append(f"{kn}-tile kernel {us:.0f} us = {2.0 * M * N * K / us / 1e6:.0f} TF/s")
    print(f"{name}: " + " | ".join(out))
