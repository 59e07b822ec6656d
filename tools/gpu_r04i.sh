#!/bin/bash
# round 4, session i: GPU suite with the fair-sharing TAS cycles and the step path's reason records; cfg5-cycle at 50k pending with the
# 5-cycle gate, cfg5f-cycle; cfg4f with the target list in LDS; rocprofv3 kernel stats of cfg4f / cfg3f / cfg5f-cycle
O=gpurun_out/r04i; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked","split")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
Q="--no-cpu-baseline --full-run 0 --no-host-leg"
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
TMO=900 run cfg4f --workload cfg4f --steps 1 --warmup 0 $Q
run cfg5cycle --workload cfg5-cycle --steps 5 --warmup 1
run cfg5fcycle --workload cfg5f-cycle --steps 5 --warmup 1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_cfg5f -o cfg5f -- python bench.py --workload cfg5f-cycle --steps 3 --warmup 1 $Q --no-parity-gate > $O/rocprof_cfg5f.log 2>&1; ls $O/prof_cfg5f | head
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_cfg3f -o cfg3f -- python bench.py --workload cfg3f --steps 10 $Q --no-parity-gate > $O/rocprof_cfg3f.log 2>&1; ls $O/prof_cfg3f | head
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_cfg4f -o cfg4f -- python bench.py --workload cfg4f --steps 1 --warmup 0 $Q --no-parity-gate > $O/rocprof_cfg4f.log 2>&1; ls $O/prof_cfg4f | head
