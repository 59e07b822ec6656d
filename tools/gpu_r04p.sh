#!/bin/bash
# round 4, session p: rocprofv3 kernel stats + PMC passes of the round's final build (cfg3, cfg3f, cfg4c, cfg5-cycle, cfg4f), the TAS segment timers,
# the default bench line with the slowest step's kernel phases
O=gpurun_out/r04p; mkdir -p $O
timeout 300 python tools/prof_tas_cycle.py > $O/prof_tas_cycle.txt 2>&1; grep "sum of\|kernel ms\|prefetched" $O/prof_tas_cycle.txt
timeout 600 python bench.py --no-cpu-baseline --full-run 0 --no-host-leg > $O/bench_default_100.json 2> $O/bench_default_100.err; python -c "
import json; d=json.load(open('$O/bench_default_100.json')); print({k:d.get(k) for k in ('ms_per_step','p50_cycle_ms','p99_cycle_ms','max_cycle_ms')})"
PROF_WORKLOADS="cfg3 cfg5-cycle cfg3f cfg4c cfg4f" bash tools/prof_round.sh r04p none profiles 2>&1 | tail -8
echo done
