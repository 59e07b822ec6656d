#!/bin/bash
# round 4, session s: A/B of the TAS unit: flat_* (main of session o) / ds_* / ds_* + split-phase copy
O=gpurun_out/r04s; mkdir -p $O
for v in "" _ldsds _early; do
  KQ_ENGINE_LIB=$PWD/kueue_amd/libkq_engine$v.so timeout 300 python bench.py --workload cfg5-cycle --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg5cycle$v.json 2> $O/bench_cfg5cycle$v.err
  python -c "
import json; d=json.load(open('$O/bench_cfg5cycle$v.json')); print('$v', {k:d.get(k) for k in ('value','ms_per_step','parity_checked')})"
done
KQ_ENGINE_LIB=$PWD/kueue_amd/libkq_engine_early.so timeout 600 python -m pytest tests/test_tas_cycle_engine.py -m gpu -q -x > $O/tests_early.log 2>&1; tail -1 $O/tests_early.log
KQ_ENGINE_LIB=$PWD/kueue_amd/libkq_engine_early.so KQ_TAS_COOP_MIN=1 timeout 300 python tools/fuzz_tas_cycle.py 0 300 hip 2>&1 | tail -1
