#!/bin/bash
# Build an instrumented variant of the library: tools/build_timing_lib.sh <out.so> [extra -D flags for gru_persistent.hip]
set -e
cd "$(dirname "$0")/../nejm-brain-to-text_amd/csrc"
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DB2T_TIMING "$@" -x hip -c gru_persistent.hip -o /tmp/gru_persistent_timing_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DB2T_TIMING "$@" -x hip -c gru_pipeline.hip -o /tmp/gru_pipeline_timing_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DB2T_TIMING "$@" -x hip -c gru_stack.hip -o /tmp/gru_stack_timing_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" gemm.o gemm_bf16.o elementwise.o gru.o /tmp/gru_persistent_timing_$$.o /tmp/gru_pipeline_timing_$$.o /tmp/gru_stack_timing_$$.o gru_granule.o ctc.o optimizer.o decode.o beam.o core.o
rm -f /tmp/gru_persistent_timing_$$.o /tmp/gru_pipeline_timing_$$.o /tmp/gru_stack_timing_$$.o
