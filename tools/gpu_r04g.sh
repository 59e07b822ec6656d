#!/bin/bash
# round 4, session g: process-only search profile of cfg4f (nominate counters to the sink), cfg4f-split with the fair target budget, GPU suite
O=gpurun_out/r04g; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked","split")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
KQ_PROF_SKIP_NOMINATE=1 timeout 400 python tools/prof_fair.py 1000 > $O/prof_fair_cfg4f_process_only.txt 2>&1; grep -v " 0 cycles" $O/prof_fair_cfg4f_process_only.txt | head -50
TMO=900 run cfg4fsplit --workload cfg4f-split --steps 1 --warmup 0
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
