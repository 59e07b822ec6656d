cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03r
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name"; cut -c1-200 $O/bench_$name.json; }
TMO=900 run cfg3
run cfg3f --workload cfg3f --steps 30 --full-run 0 --no-host-leg
run cfg4c --workload cfg4c --steps 3 --warmup 1 --cpu-seconds 5
run cfg5cycle --workload cfg5-cycle --steps 10 --warmup 2 --cpu-seconds 5
TMO=900 run cfg4f --workload cfg4f --steps 1 --warmup 0 --cpu-seconds 5 --no-host-leg
python - <<PY
import json
d=json.load(open("$O/bench_cfg3.json")); print(d["value"], d["full_run"]["complete"], d["full_run"]["cycles"], d["full_run"]["workloads_decided"], d["full_run"]["still_active"], d["full_run"]["decisions_per_s"])
PY
