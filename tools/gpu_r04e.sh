#!/bin/bash
# round 4, session e: Fs in registers + the fair iterator's state in LDS: GPU suite, cfg4f with the batch bits, cfg3f with / without the LDS iterator
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
Q="--no-cpu-baseline --full-run 0 --no-host-leg"
TMO=900 run cfg4f --workload cfg4f --steps 1 --warmup 0 $Q
KQ_FS_BATCH=13 TMO=900 run cfg4f_b13 --workload cfg4f --steps 1 --warmup 0 $Q --no-parity-gate
KQ_FS_BATCH=0 TMO=900 run cfg4f_b0 --workload cfg4f --steps 1 --warmup 0 $Q --no-parity-gate
run cfg3f --workload cfg3f --steps 30 $Q
KQ_FS_ITER_LDS=0 run cfg3f_noiter --workload cfg3f --steps 30 $Q --no-parity-gate
