#!/bin/bash
# round 4, session t (the last of the round, 21 GPU-minutes left): the round's final build — the whole GPU suite incl. the node-replacement /
# exclusion-statistics tests, the default bench line exactly as the driver runs it, cfg5-cycle, cfg5 (with its parity gate), cfg5f-cycle
O=gpurun_out/r04t; mkdir -p $O
show() { python -c "
import json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$1', {k:d.get(k) for k in ('value','ms_per_step','p99_cycle_ms','max_cycle_ms','kernel_ms_per_cycle','parity_checked')})"; }
timeout 540 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 150 python bench.py --steps 20 --warmup 5 > $O/bench_default_driver.json 2> $O/bench_default_driver.err; show $O/bench_default_driver.json
timeout 150 python bench.py --workload cfg5-cycle --steps 5 --warmup 1 > $O/bench_cfg5-cycle.json 2> $O/bench_cfg5-cycle.err; show $O/bench_cfg5-cycle.json
timeout 100 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; show $O/bench_cfg5.json
timeout 150 python bench.py --workload cfg5f-cycle --steps 5 --warmup 1 > $O/bench_cfg5f-cycle.json 2> $O/bench_cfg5f-cycle.err; show $O/bench_cfg5f-cycle.json
timeout 240 python bench.py > $O/bench_default.json 2> $O/bench_default.err; show $O/bench_default.json
grep -l "Traceback" $O/*.log $O/*.err 2>/dev/null
echo done
