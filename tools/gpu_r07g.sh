#!/bin/bash
# r07g: the final build of round 6 on the MI355X - whole GPU suite, smoke, every bench line, rocprofv3 stats + PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
tools/gpu_session.sh r07g tests smoke
PROF_WORKLOADS="cfg3 cfg2 cfg3f cfg4c cfg5 cfg5-cycle cfg4f" tools/prof_round.sh r07g benches profiles > gpurun_out/r07g/prof.log 2>&1
tail -5 gpurun_out/r07g/prof.log
