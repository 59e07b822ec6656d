#!/bin/bash
# round 4, session d: Fs back in registers (force-inlined batch functions): cfg4f with the batch bits on / without the second-strategy batch / off
O=gpurun_out/r04d; mkdir -p $O
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
run cfg4c --workload cfg4c --steps 3 --warmup 1 $Q
KQ_CS_LAZY=2 run cfg4c_lazy2 --workload cfg4c --steps 3 --warmup 1 $Q --no-parity-gate
timeout 300 python tools/prof_process.py 3 fair > $O/prof_process_cfg3f.txt 2>&1; cat $O/prof_process_cfg3f.txt | head -40
