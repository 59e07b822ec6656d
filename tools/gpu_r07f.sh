#!/bin/bash
# r07f: the build with two count arrays for a private phase 1 without a leader - TAS GPU tests, the TAS bench lines, PMC passes of cfg5-cycle
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_session.sh r07f "tests:tas or TAS or closed_loop" bench:cfg5-cycle:"--steps 20 --warmup 4" bench:cfg5:"--steps 5 --warmup 1" bench:cfg5f-cycle:"--steps 10 --warmup 2"
PROF_WORKLOADS="cfg5-cycle" tools/prof_round.sh r07f none profiles > gpurun_out/r07f/prof.log 2>&1
tail -3 gpurun_out/r07f/prof.log
