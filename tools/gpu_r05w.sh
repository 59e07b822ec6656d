#!/bin/bash
# r05w: the final build's TAS lines once more + rocprofv3 kernel stats and the PMC passes of the closed TAS loop and the TAS batch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05w; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name"; cut -c1-260 $O/bench_$name.json; }
run default_driver --steps 20 --warmup 5
run cfg5 --workload cfg5 --steps 5 --warmup 1
run cfg5cycle --workload cfg5-cycle --steps 20 --warmup 4
run cfg5fcycle --workload cfg5f-cycle --steps 10 --warmup 2
run cfg5cycle_open --workload cfg5-cycle --open-loop --steps 10 --warmup 2 --cpu-seconds 5
run cfg5split --workload cfg5-split --steps 5 --warmup 1
PROF_WORKLOADS="cfg5-cycle cfg5" bash tools/prof_round.sh r05w none profiles 2>&1 | tail -n 4
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
