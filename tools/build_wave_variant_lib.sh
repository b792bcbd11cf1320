#!/bin/bash
# A variant of the library with extra -D switches on csrc/gru_wave.hip (diagnosis builds): tools/build_wave_variant_lib.sh <out.so> -DX ...
set -e
cd "$(dirname "$0")/../nejm-brain-to-text_amd/csrc"
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include "$@" -x hip -c gru_wave.hip -o /tmp/gru_wave_var_$$.o
objs=$(ls *.o | grep -v '^gru_wave.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs /tmp/gru_wave_var_$$.o
rm -f /tmp/gru_wave_var_$$.o
