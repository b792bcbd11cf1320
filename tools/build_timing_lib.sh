#!/bin/bash
# Build an instrumented variant of the library (per-phase cycle counters in the forward sweep, csrc/gru_persistent.hip):
#   tools/build_timing_lib.sh <out.so> [extra -D flags for gru_persistent.hip]     (run __graft_entry__.build() first)
# Select it with B2T_LIB=<out.so>; tools/sweep_anatomy.py (one sweep alone) and tools/step_anatomy.py (inside the step) read it.
set -e
cd "$(dirname "$0")/../nejm-brain-to-text_amd/csrc"
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DB2T_TIMING "$@" -x hip -c gru_persistent.hip -o /tmp/gru_persistent_timing_$$.o
objs=$(ls *.o | grep -v '^gru_persistent.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs /tmp/gru_persistent_timing_$$.o
rm -f /tmp/gru_persistent_timing_$$.o
