#!/bin/bash
# ThreadSanitizer over the threaded determinisation of Rescore() (csrc/graphc.cpp, DetRescore::run with helper threads): the host
# half of the library built host-only with -fsanitize=thread, driven from Python under the sanitizer's runtime.
# Needs gpurun_out/lattices.npz (attic/dump_lattice.py on a GPU box).  Round 5: no report in any configuration below.
set -e
cd "$(dirname "$0")/.."
CL=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname "$($CL -print-libgcc-file-name --rtlib=compiler-rt)")/libclang_rt.tsan-x86_64.so
[ -f "$RT" ] || RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.tsan-x86_64.so" | head -1)
C=nejm-brain-to-text_amd/csrc
$CL -x hip --offload-host-only --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -shared -Iinclude -I$C $C/graphc.cpp $C/lattice.cpp $C/core.cpp -o /tmp/libb2t_tsan.so
for cfg in "0 6 192" "0 4 16" "23 3 64"; do
  set -- $cfg
  U=$1 REPS=3 B2T_RESCORE_THREADS=$2 B2T_RESCORE_MIN_BATCH=$3 B2T_LAT_TIMING=1 LD_PRELOAD=$RT TSAN_OPTIONS="report_signal_unsafe=0 exitcode=0" \
    python tools/r5_rescore_ab.py /tmp/libb2t_tsan.so 2>&1 | grep -E "WARNING|data race|det: |SUMMARY|libb2t_tsan" | sed 's/.*det: /det: /' | sort | uniq -c
done
