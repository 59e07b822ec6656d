#!/bin/bash
# experiment builds of the speculative process kernel against the engine object of the last full build:
#   tools/build_spec_variant.sh <suffix> [extra hipcc flags]   ->  kueue_amd/libkq_engine_<suffix>.so
set -e
cd "$(dirname "$0")/.."
S=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -Rpass-analysis=kernel-resource-usage -c -o build/kq_spec_kernel_$S.o kueue_amd/csrc/kq_spec_kernel.hip 2>&1 | grep -E "error|VGPRs:|Scratch|VGPRs Spill" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kueue_amd/libkq_engine_$S.so build/kq_engine.o build/kq_spec_kernel_$S.o build/kq_rows_kernel.o build/kq_tas_cycle_kernel.o build/kq_tas_cycle_kernel_bal.o build/kq_tas_bal_kernel.o build/kq_group.o -ldl -lpthread
