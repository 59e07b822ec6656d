#!/bin/bash
# r07i: k_usage_cols over whole 64-byte lines - whole GPU suite, smoke, default line (driver form + 100 steps), cfg2, cfg3f, rocprofv3 + PMC passes of cfg3
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_session.sh r07i tests smoke bench:cfg3:"--steps 20 --warmup 5" bench bench:cfg2:"--full-run 0" bench:cfg3f:"--steps 30 --full-run 0 --no-host-leg"
PROF_WORKLOADS="cfg3" tools/prof_round.sh r07i none profiles > gpurun_out/r07i/prof.log 2>&1
tail -3 gpurun_out/r07i/prof.log
