#!/bin/bash
# kueue_amd/libkq_engine_prof.so = the engine with the in-kernel segment timers (-DKQ_PROF): tools/prof_process.py, tools/prof_lean.py, tools/prof_fair.py
# Compiles from a snapshot of the sources (build/_src_prof): hipcc reads a translation unit twice (device pass, host pass), and an edit
# in between gives a library whose host side launches kernels its device side does not have.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
rm -rf build/_src_prof; mkdir -p build/_src_prof/kueue_amd
cp -r kueue_amd/csrc build/_src_prof/kueue_amd/csrc; cp -r include build/_src_prof/include
S=build/_src_prof/kueue_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DKQ_PROF"
/opt/rocm/bin/hipcc $F -c -o build/kq_engine_prof.o $S/kq_engine.hip &
/opt/rocm/bin/hipcc $F -c -o build/kq_tas_cycle_kernel_prof.o $S/kq_tas_cycle_kernel.hip &
/opt/rocm/bin/hipcc $F -c -o build/kq_spec_kernel_prof2.o $S/kq_spec_kernel.hip
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kueue_amd/libkq_engine_prof.so build/kq_engine_prof.o build/kq_spec_kernel_prof2.o build/kq_tas_cycle_kernel_prof.o build/kq_rows_kernel.o build/kq_group.o build/kq_tas_cycle_kernel_bal.o build/kq_tas_bal_kernel.o -ldl -lpthread  # (the row kernels carry no timers: the ordinary object)
