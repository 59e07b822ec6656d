#!/bin/bash
# Measurement pass of a round on the GPU box: benches (JSON lines) + rocprofv3 kernel stats + PMC passes (FETCH_SIZE / WRITE_SIZE in
# separate runs, never combined with other trace domains). Everything lands under gpurun_out/$1/.
#   tools/prof_round.sh r03 [benches] [profiles]      (default: both)
TAG=${1:-r03}
WHAT="${2:-benches} ${3:-profiles}"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name"; cut -c1-700 $O/bench_$name.json; }
if [[ $WHAT == *benches* ]]; then
  TMO=900 run cfg3
  run cfg3_sync --loop sync --no-cpu-baseline --full-run 0 --no-host-leg
  run cfg2 --workload cfg2 --full-run 0
  run cfg3f --workload cfg3f --steps 30 --full-run 0 --no-host-leg
  run cfg3b --workload cfg3-batch --steps 20 --warmup 2
  run cfg3split --workload cfg3-split --steps 50
  run cfg4csplit --workload cfg4c-split --steps 3 --warmup 1
  run cfg4c_open --workload cfg4c --open-loop --steps 3 --warmup 1 --cpu-seconds 5
  run cfg5 --workload cfg5 --steps 5 --warmup 1
  run cfg5split --workload cfg5-split --steps 5 --warmup 1
  run cfg5cycle --workload cfg5-cycle --steps 20 --warmup 4
  run cfg5cycle_open --workload cfg5-cycle --open-loop --steps 10 --warmup 2 --cpu-seconds 5
  run cfg5fcycle --workload cfg5f-cycle --steps 10 --warmup 2
  run cfg3group --workload cfg3-group --steps 20 --warmup 5
  run cfg4cgroup --workload cfg4c-group --steps 5 --warmup 1
  run cfg4fgroup --workload cfg4f-group --steps 10 --warmup 2
  run default_driver --steps 20 --warmup 5
  TMO=900 run cfg4f_open --workload cfg4f --open-loop --steps 2 --warmup 1 --cpu-seconds 5 --no-host-leg
  # BASELINE configs[3] as the closed loop (round 6): both start states, classical and fair
  TMO=1500 run cfg4c_feasible --workload cfg4c --start feasible --steps 60 --series-cycles 20 --cpu-seconds 10
  TMO=1500 run cfg4c --workload cfg4c --steps 40 --series-cycles 6 --cpu-seconds 10
  TMO=1500 run cfg4f_feasible --workload cfg4f --start feasible --steps 40 --series-cycles 14 --cpu-seconds 10
  TMO=1500 run cfg4f --workload cfg4f --steps 16 --series-cycles 6 --cpu-seconds 10
fi
if [[ $WHAT == *profiles* ]]; then
  cd /tmp && export TMPDIR=/tmp
  Q="--no-cpu-baseline --no-parity-gate --full-run 0 --no-host-leg"
  declare -A CMD
  CMD[cfg3]="python $R/bench.py $Q"
  CMD[cfg3f]="python $R/bench.py --workload cfg3f --steps 20 $Q"
  CMD[cfg4c]="python $R/bench.py --workload cfg4c --start feasible --steps 24 --warmup 4 --series-cycles 0 --no-cpu-baseline --no-parity-gate"
  CMD[cfg4f]="python $R/bench.py --workload cfg4f --start feasible --steps 12 --warmup 4 --series-cycles 0 --no-cpu-baseline --no-parity-gate"
  CMD[cfg2]="python $R/bench.py --workload cfg2 $Q"
  CMD[cfg5]="python $R/bench.py --workload cfg5 --steps 5 --warmup 1 --no-cpu-baseline"
  CMD[cfg5-cycle]="python $R/bench.py --workload cfg5-cycle --steps 5 --warmup 1 --no-cpu-baseline --no-parity-gate"
  for W in ${PROF_WORKLOADS:-cfg3 cfg3f cfg4c cfg5 cfg4f}; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_${W}_stats -- ${CMD[$W]} > $O/p_${W}_stats.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_${W}_fetch -- ${CMD[$W]} > $O/p_${W}_fetch.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_${W}_write -- ${CMD[$W]} > $O/p_${W}_write.log 2>&1
    echo "== profiled $W"
  done
  cd $R
  python tools/cycle_timeline.py $O/p_cfg3_stats > $O/cfg3_timeline.txt 2>&1
fi
tail -n 2 $O/*.err 2>/dev/null | grep -v amdgpu.ids | grep -v "^$" | head -40
