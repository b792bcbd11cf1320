#!/bin/bash
# Instrumented variant of the library for the layer wavefront (csrc/gru_wave.hip, -DB2T_WAVE_TIMING: cycles per step and phase of
# slice 0 / wave 0 of every layer).  tools/build_wave_timing_lib.sh <out.so>; select with B2T_LIB=<out.so>; tools/r6_wave_probe.py prints it.
set -e
cd "$(dirname "$0")/../nejm-brain-to-text_amd/csrc"
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DB2T_WAVE_TIMING "$@" -x hip -c gru_wave.hip -o /tmp/gru_wave_timing_$$.o
objs=$(ls *.o | grep -v '^gru_wave.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs /tmp/gru_wave_timing_$$.o
rm -f /tmp/gru_wave_timing_$$.o
