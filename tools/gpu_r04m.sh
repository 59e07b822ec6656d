#!/bin/bash
# round 4, session m: k_process_tas with cooperative sweeps, header prefetch, asynchronous class-table patch, LDS request block: TAS GPU tests + random cycles through the
# C ABI, segment timers with the LDS state on / off, cfg5-cycle bench lines on / off
O=gpurun_out/r04m; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked")})
except Exception as e: print("no json", e)
PY
tail -2 $O/bench_$name.err | grep -v amdgpu.ids
}
Q="--no-cpu-baseline"
timeout 900 python -m pytest tests/test_tas_cycle_engine.py tests/test_gpu_tas.py -m gpu -q -x > $O/gpu_tas_tests.log 2>&1; tail -2 $O/gpu_tas_tests.log
timeout 300 python tools/fuzz_tas_cycle.py 0 400 hip > $O/fuzz_hip.txt 2>&1; tail -1 $O/fuzz_hip.txt
KQ_TAS_COOP_MIN=1 timeout 300 python tools/fuzz_tas_cycle.py 400 800 hip > $O/fuzz_hip_coop1.txt 2>&1; tail -1 $O/fuzz_hip_coop1.txt
timeout 300 python tools/prof_tas_cycle.py > $O/prof_tas_cycle_lds.txt 2>&1; cat $O/prof_tas_cycle_lds.txt | grep -v amdgpu.ids
KQ_TAS_LDS_OFF=1 timeout 300 python tools/prof_tas_cycle.py > $O/prof_tas_cycle_nolds.txt 2>&1; grep "recomputation (\|placement (\|sum of\|kernel ms\|findLevel" $O/prof_tas_cycle_nolds.txt
run cfg5cycle --workload cfg5-cycle --steps 5 --warmup 1 $Q
KQ_TAS_LDS_OFF=1 run cfg5cycle_nolds --workload cfg5-cycle --steps 5 --warmup 1 $Q
echo done
