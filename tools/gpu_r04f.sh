#!/bin/bash
# round 4, session f: the batch only in losing streaks; adaptive fill-back; profiles of the fair cycle and of the fair iterator
O=gpurun_out/r04f; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked","split")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
Q="--no-cpu-baseline --full-run 0 --no-host-leg"
TMO=900 run cfg4f --workload cfg4f --steps 1 --warmup 0 $Q
KQ_FS_BATCH=31 TMO=900 run cfg4f_b31 --workload cfg4f --steps 1 --warmup 0 $Q --no-parity-gate
run cfg3f --workload cfg3f --steps 30 $Q
timeout 400 python tools/prof_fair.py 1000 > $O/prof_fair_cfg4f.txt 2>&1; grep -v " 0 cycles" $O/prof_fair_cfg4f.txt | head -50
timeout 300 python tools/prof_process.py 3 fair > $O/prof_process_cfg3f.txt 2>&1; grep "fair:\|kernel ms" $O/prof_process_cfg3f.txt
TMO=900 run cfg4fsplit --workload cfg4f-split --steps 1 --warmup 0
