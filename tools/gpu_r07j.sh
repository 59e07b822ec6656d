#!/bin/bash
# r07j: bench.py cfg5-cycle / cfg5f-cycle with the ENGINE-driven closed-loop trajectory (+ node failures)
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_session.sh r07j bench:cfg5-cycle:"--steps 20 --warmup 4" bench:cfg5f-cycle:"--steps 10 --warmup 2" bench:cfg5-cycle:"--steps 12 --warmup 2 --node-failures 16 --no-cpu-baseline"
