#!/bin/bash
# round 4, session n: k_process_tas with the per-tree state in LDS; TAS GPU tests, cfg5-cycle / cfg5f-cycle / cfg5 bench lines
O=gpurun_out/r04n; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
Q="--no-cpu-baseline"
timeout 900 python -m pytest tests/test_tas_cycle_engine.py tests/test_gpu_tas.py tests/test_zz_gpu_tas_admit.py tests/test_zz_gpu_tas_multilayer.py -m gpu -q -x > $O/gpu_tas_tests.log 2>&1; tail -2 $O/gpu_tas_tests.log
run cfg5cycle --workload cfg5-cycle --steps 5 --warmup 1 $Q
run cfg5fcycle --workload cfg5f-cycle --steps 5 --warmup 1 $Q
run cfg5 --workload cfg5 $Q
echo done
