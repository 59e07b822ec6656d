#!/bin/bash
# an A/B build of the TAS cycle's translation unit: tools/build_tas_variant.sh <name> <extra hipcc flags...> -> kueue_amd/libkq_engine_<name>.so (KQ_ENGINE_LIB selects it)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
D=build/_src_$name
rm -rf $D; mkdir -p $D/kueue_amd
cp -r kueue_amd/csrc $D/kueue_amd/csrc; cp -r include $D/include
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c -o build/kq_tas_cycle_kernel_$name.o $D/kueue_amd/csrc/kq_tas_cycle_kernel.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o kueue_amd/libkq_engine_$name.so build/kq_engine.o build/kq_spec_kernel.o build/kq_tas_cycle_kernel_$name.o build/kq_rows_kernel.o build/kq_group.o -ldl -lpthread
