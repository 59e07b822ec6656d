#!/bin/bash
# Round-end measurement pass: default bench (cfg3, closed loop) and cfg5 under rocprofv3 (stats + FETCH_SIZE + WRITE_SIZE passes).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for W in ${@:-cfg3 cfg5}; do
  if [ $W = cfg3 ]; then CMD="python $R/bench.py --no-cpu-baseline"; else CMD="python $R/bench.py --workload cfg5 --steps 5 --warmup 1 --no-cpu-baseline"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_${W}_stats -- $CMD > $R/gpurun_out/p_${W}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_${W}_fetch -- $CMD > $R/gpurun_out/p_${W}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_${W}_write -- $CMD > $R/gpurun_out/p_${W}_write.log 2>&1
done
