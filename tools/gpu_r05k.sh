#!/bin/bash
# r05k: the closed TAS loop with (default) and without (KQ_TAS_EMPTY_TABLES_OFF=1) the empty-cluster class tables; the em_ps reuse is in both
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05k; mkdir -p $O
timeout 300 python tools/prof_tas_closed.py > $O/prof_tas_closed.txt 2>&1
KQ_TAS_EMPTY_TABLES_OFF=1 timeout 300 python tools/prof_tas_closed.py > $O/prof_tas_closed_notab.txt 2>&1
timeout 300 python bench.py --workload cfg5-cycle --steps 20 --warmup 4 > $O/bench_cfg5_cycle.json 2> $O/bench_cfg5_cycle.err
KQ_TAS_EMPTY_TABLES_OFF=1 timeout 300 python bench.py --workload cfg5-cycle --steps 20 --warmup 4 > $O/bench_cfg5_cycle_notab.json 2> $O/bench_cfg5_cycle_notab.err
timeout 600 python -m pytest tests/test_tas_closed_loop.py tests/test_tas_cycle_engine.py -m gpu -x -q -n 2 > $O/pytest_tas.txt 2>&1
tail -n 3 $O/pytest_tas.txt
grep -h "sum of the entry\|kernel ms\|placement (t_workload)\|phase 1\|recomputation (get" $O/prof_tas_closed.txt $O/prof_tas_closed_notab.txt
cat $O/bench_cfg5_cycle.json $O/bench_cfg5_cycle_notab.json | cut -c1-400
