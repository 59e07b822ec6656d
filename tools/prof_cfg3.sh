#!/bin/bash
# rocprofv3 passes for the default bench (cfg3, closed loop): kernel-trace stats, then FETCH_SIZE and WRITE_SIZE PMC runs.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3_stats -- $CMD > $R/gpurun_out/prof3_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof3_fetch -- $CMD > $R/gpurun_out/prof3_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof3_write -- $CMD > $R/gpurun_out/prof3_write.log 2>&1
