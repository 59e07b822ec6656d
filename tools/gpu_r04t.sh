#!/bin/bash
# round 4, session t: the round's final build — the whole GPU suite, the default bench line exactly as the driver runs it (20 steps, 5 warm-up) and with
# no flags, cfg5-cycle / cfg5f-cycle / cfg5, the TAS segment timers, rocprofv3 kernel stats + PMC passes of cfg5-cycle
O=gpurun_out/r04t; mkdir -p $O
show() { python -c "
import json; d=json.load(open('$1')); print({k:d.get(k) for k in ('value','ms_per_step','p50_cycle_ms','p99_cycle_ms','max_cycle_ms','kernel_ms_per_cycle','parity_checked')})"; }
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default_driver.json 2> $O/bench_default_driver.err; show $O/bench_default_driver.json
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; show $O/bench_default.json
for w in cfg5-cycle cfg5f-cycle; do timeout 600 python bench.py --workload $w --steps 5 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err; show $O/bench_$w.json; done
timeout 300 python bench.py --workload cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; show $O/bench_cfg5.json
timeout 300 python tools/prof_tas_cycle.py > $O/prof_tas_cycle.txt 2>&1; grep "sum of\|kernel ms\|prefetched" $O/prof_tas_cycle.txt
PROF_WORKLOADS="cfg5-cycle" bash tools/prof_round.sh r04t none profiles 2>&1 | tail -3
grep -l "Traceback" $O/*.log $O/*.err 2>/dev/null
echo done
