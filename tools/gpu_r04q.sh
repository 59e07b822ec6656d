#!/bin/bash
# round 4, session q: the default bench line with the interpreter's collector frozen for the timed region (100 and 20 steps), cfg2; rocprofv3 kernel
# stats + PMC passes of cfg4c / cfg4f on the final build (session p lost them to a bench.py typo)
O=gpurun_out/r04q; mkdir -p $O
show() { python -c "
import json; d=json.load(open('$1')); print({k:d.get(k) for k in ('value','ms_per_step','p50_cycle_ms','p99_cycle_ms','max_cycle_ms','issue_to_readable_ms')})"; }
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; show $O/bench_default.json
timeout 600 python bench.py --steps 20 --no-cpu-baseline --full-run 0 --no-host-leg --no-parity-gate > $O/bench_default_20.json 2> $O/bench_default_20.err; show $O/bench_default_20.json
timeout 600 python bench.py --workload cfg2 --full-run 0 --no-cpu-baseline > $O/bench_cfg2.json 2> $O/bench_cfg2.err; show $O/bench_cfg2.json
PROF_WORKLOADS="cfg4c cfg4f" bash tools/prof_round.sh r04q none profiles 2>&1 | tail -4
grep -l "Traceback" $O/*.log $O/*.err 2>/dev/null
echo done
