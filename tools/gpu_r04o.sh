#!/bin/bash
# round 4, session o: the whole GPU suite on the build with the gathered quota steps, slices in the resident pending set and the 16-level /
# 32-resource TAS limits; cfg5-cycle / cfg5f-cycle / cfg4c / default bench lines
O=gpurun_out/r04o; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
Q="--no-cpu-baseline"
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
run cfg5cycle --workload cfg5-cycle --steps 5 --warmup 1 $Q
run cfg5fcycle --workload cfg5f-cycle --steps 5 --warmup 1 $Q
run cfg4c --workload cfg4c --steps 5 $Q --full-run 0 --no-host-leg
TMO=900 run default
echo done
