#!/bin/bash
# libkq_engine_specprof.so = the engine's objects + the spec kernel built with section timers (tools/prof_spec.py)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DKQ_SPEC_PROF -c -o build/kq_spec_kernel_prof.o kueue_amd/csrc/kq_spec_kernel.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kueue_amd/libkq_engine_specprof.so build/kq_engine.o build/kq_spec_kernel_prof.o build/kq_rows_kernel.o build/kq_tas_cycle_kernel.o build/kq_tas_cycle_kernel_bal.o build/kq_tas_bal_kernel.o build/kq_group.o -ldl -lpthread
