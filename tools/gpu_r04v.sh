#!/bin/bash
# round 4, session v: after the wave-aggregated counters of k_tas_excl — the GPU suite, the node-replacement measurement at the cfg 5 topology
# with its rocprofv3 kernel stats, rocprofv3 kernel stats + PMC passes of the final build for cfg 3 (default line) and cfg 5
O=gpurun_out/r04v; mkdir -p $O
timeout 540 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 300 python tools/bench_tas_replacement.py 2000 > $O/bench_tas_replacement.json 2> $O/bench_tas_replacement.err; cat $O/bench_tas_replacement.json; tail -3 $O/bench_tas_replacement.err
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/tools/bench_tas_replacement.py 2000 > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" && cp "$f" $O/repl_kernel_stats.csv
rm -rf $O/prof
PROF_WORKLOADS="cfg3 cfg5" bash tools/prof_round.sh r04v none profiles 2>&1 | tail -4
du -sh $O | tail -1
echo done
