#!/bin/bash
# r07h: the build with node-feasibility masks in the TAS cycle - whole GPU suite, smoke, the default line as the driver runs it, the TAS lines
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_session.sh r07h tests smoke bench:cfg3:"--steps 20 --warmup 5" bench:cfg5-cycle:"--steps 20 --warmup 4" bench:cfg5:"--steps 5 --warmup 1" bench:cfg5f-cycle:"--steps 10 --warmup 2"
