#!/bin/bash
# r07m: HIP fuzz of TAS cycles (first pass, second pass, balanced) beyond the pinned seeds on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_session.sh r07m "py:tools/fuzz_tas_cycle.py 20000 22000 hip" "py:tools/fuzz_tas_cycle.py 30000 31000 hip second" "py:tools/fuzz_balanced.py 5000 5400 hip"
