#!/bin/bash
# round 4, session u: the node-replacement entry points at the cfg 5 topology (parity against the oracle on 2000 workloads, then timed),
# rocprofv3 kernel stats of the same run, the replacement tests once more
O=gpurun_out/r04u; mkdir -p $O
timeout 300 python tools/bench_tas_replacement.py 2000 > $O/bench_tas_replacement.json 2> $O/bench_tas_replacement.err; cat $O/bench_tas_replacement.json; tail -3 $O/bench_tas_replacement.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/bench_tas_replacement.py 500 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" && cp "$f" $O/repl_kernel_stats.csv; find $O/prof -type f | head -5
rm -rf $O/prof
echo done
