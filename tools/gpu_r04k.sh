#!/bin/bash
# round 4, session k: computeDRS as (ClusterQueue, level) items (cfg3f), commit of an evaluated removal (cfg4f), process-only profiles,
# the default bench line
O=gpurun_out/r04k; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked","split")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
Q="--no-cpu-baseline --full-run 0 --no-host-leg"
run cfg3f --workload cfg3f --steps 30 $Q
TMO=900 run cfg4f --workload cfg4f --steps 1 --warmup 0 $Q
run cfg4c --workload cfg4c --steps 5 $Q
timeout 300 python tools/prof_process.py 3 fair > $O/prof_process_cfg3f.txt 2>&1; grep "fair:\|kernel ms" $O/prof_process_cfg3f.txt
KQ_PROF_SKIP_NOMINATE=1 timeout 400 python tools/prof_fair.py 1000 > $O/prof_fair_cfg4f_process_only.txt 2>&1; grep "search:\|lds search\|recompute" $O/prof_fair_cfg4f_process_only.txt | grep -v " 0 cycles"
KQ_PROF_SKIP_NOMINATE=1 timeout 300 python tools/prof_cfg4c.py > $O/prof_cfg4c_process_only.txt 2>&1; grep -v " 0 cycles" $O/prof_cfg4c_process_only.txt | head -40
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_population.py tests/test_fair_lds_search.py tests/test_pending_step.py -m gpu -q > $O/gpu_tests_subset.log 2>&1; tail -3 $O/gpu_tests_subset.log
TMO=900 run default
echo done
