#!/bin/bash
# kueue_amd/libkq_engine_prof.so = the engine with the in-kernel segment timers (-DKQ_PROF): tools/prof_process.py, tools/prof_lean.py, tools/prof_fair.py
set -e
cd "$(dirname "$0")/.."
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DKQ_PROF -c -o build/kq_engine_prof.o kueue_amd/csrc/kq_engine.hip &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DKQ_PROF -c -o build/kq_tas_cycle_kernel_prof.o kueue_amd/csrc/kq_tas_cycle_kernel.hip &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DKQ_PROF -c -o build/kq_spec_kernel_prof2.o kueue_amd/csrc/kq_spec_kernel.hip
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kueue_amd/libkq_engine_prof.so build/kq_engine_prof.o build/kq_spec_kernel_prof2.o build/kq_tas_cycle_kernel_prof.o build/kq_rows_kernel.o   # (the row kernels carry no timers: the ordinary object)
