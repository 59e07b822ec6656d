#!/bin/bash
# r05v: HIP fuzz beyond the pinned seeds: second-pass cycles, balanced batches and cycles
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05v; mkdir -p $O
timeout 400 python tools/fuzz_tas_cycle.py 7000 8200 hip second > $O/fuzz_second_hip.txt 2>&1; tail -n 2 $O/fuzz_second_hip.txt
timeout 400 python tools/fuzz_balanced.py 3000 4000 hip > $O/fuzz_balanced_hip.txt 2>&1; tail -n 2 $O/fuzz_balanced_hip.txt
timeout 300 python tools/fuzz_tas_cycle.py 100000 100800 hip > $O/fuzz_tas_cycle_hip.txt 2>&1; tail -n 1 $O/fuzz_tas_cycle_hip.txt
