#!/bin/bash
# r05x: updateMode on the wave + cached WorkloadsTopologyRequests: the closed TAS loop and its tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05x; mkdir -p $O
timeout 300 python bench.py --workload cfg5-cycle --steps 20 --warmup 4 > $O/bench_cfg5_cycle.json 2> $O/bench_cfg5_cycle.err
timeout 300 python bench.py --workload cfg5f-cycle --steps 10 --warmup 2 > $O/bench_cfg5f_cycle.json 2> $O/bench_cfg5f_cycle.err
timeout 300 python bench.py --workload cfg5-cycle --node-failures 16 --steps 20 --warmup 4 > $O/bench_cfg5_cycle_failures16.json 2> $O/bench_cfg5_cycle_failures16.err
timeout 900 python -m pytest tests/test_tas_cycle_engine.py tests/test_tas_closed_loop.py -m gpu -x -q -p no:cacheprovider > $O/pytest_tas_cycle.txt 2>&1; tail -n 2 $O/pytest_tas_cycle.txt
cat $O/bench_cfg5_cycle.json $O/bench_cfg5f_cycle.json $O/bench_cfg5_cycle_failures16.json | cut -c1-250
