#!/bin/bash
# round 4, session j: GPU suite after the fair walk went into the TAS translation unit; 512 vs 256 threads for k_process_fair
O=gpurun_out/r04j; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked","split")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
Q="--no-cpu-baseline --full-run 0 --no-host-leg"
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
run cfg5fcycle --workload cfg5f-cycle --steps 5 --warmup 1
KQ_ENGINE_LIB=kueue_amd/libkq_engine_t256.so TMO=900 run cfg4f_t256 --workload cfg4f --steps 1 --warmup 0 $Q
KQ_ENGINE_LIB=kueue_amd/libkq_engine_t256.so run cfg3f_t256 --workload cfg3f --steps 30 $Q
