#!/bin/bash
# rocprofv3 passes for the cfg5 (TAS) bench: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate PMC runs.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload cfg5 --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5_stats -- $CMD > $R/gpurun_out/prof5_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof5_fetch -- $CMD > $R/gpurun_out/prof5_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof5_write -- $CMD > $R/gpurun_out/prof5_write.log 2>&1
find $R/gpurun_out/prof5_* -name "*.csv" | head -20
