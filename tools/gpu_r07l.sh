#!/bin/bash
# r07l: the final tree of round 6 - whole GPU suite, smoke, the default line as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_session.sh r07l tests smoke bench:cfg3:"--steps 20 --warmup 5"
