#!/bin/bash
# r05s: BASELINE configs[4] as a closed loop WITH node failures: second-pass heads at full size next to 1000 first-pass heads, parity over all cycles
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05s; mkdir -p $O
timeout 600 python -m pytest tests/test_tas_closed_loop.py -m gpu -x -q -p no:cacheprovider > $O/pytest_closed_loop.txt 2>&1; tail -n 2 $O/pytest_closed_loop.txt
timeout 600 python bench.py --workload cfg5-cycle --node-failures 16 --steps 20 --warmup 4 > $O/bench_cfg5_cycle_failures16.json 2> $O/bench_cfg5_cycle_failures16.err
tail -n 3 $O/bench_cfg5_cycle_failures16.err
cut -c1-300 $O/bench_cfg5_cycle_failures16.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05s/bench_cfg5_cycle_failures16.json"))
print(d["value"], d["ms_per_step"], d.get("second_pass"), d["parity"])
PY
