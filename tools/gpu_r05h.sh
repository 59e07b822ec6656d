#!/bin/bash
# round 5, session h (final build): the one-time runtime stall over a 400-step window (VERDICT r04 "weak" 9), the GPU suite under the
# poisoned + guarded allocator, a second seed of the put / patch / cycle guard fuzz
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05h; mkdir -p $O
timeout 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline --full-run 0 --no-host-leg > $O/bench_cfg3_400steps.json 2> $O/bench_cfg3_400steps.err; python - <<PY
import json
d=json.load(open("$O/bench_cfg3_400steps.json")); print({k:d.get(k) for k in ("value","ms_per_step","p50_cycle_ms","p99_cycle_ms","max_cycle_ms","issue_to_readable_ms")})
PY
KQ_GUARD=1 timeout 1200 python -m pytest tests -m gpu -q -n 2 --max-worker-restart 12 > $O/pytest_gpu_poison.txt 2>&1; echo "poison pytest rc=$?" >> $O/pytest_gpu_poison.txt; tail -n 3 $O/pytest_gpu_poison.txt
KQ_GUARD=1 timeout 300 python tools/fuzz_put_guard.py --iters 10000 --seconds 200 --seed 7 > $O/fuzz_put_guard_seed7.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz_put_guard_seed7.txt; tail -n 4 $O/fuzz_put_guard_seed7.txt
