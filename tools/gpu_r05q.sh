#!/bin/bash
# r05q: the build with the k_order_scatter fix: the r05m/r05n reproduction first, then the GPU suite in ONE process as the driver runs it
# (no xdist), smoke, the default line, the closed TAS loop, second-pass cycles on the HIP engine
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05q; mkdir -p $O
for i in 1 2 3; do timeout 300 python -m pytest tests/test_tas_cycle_engine.py -m gpu -x -q -s -p no:cacheprovider -k random_tas_cycles_gpu > $O/repro_$i.txt 2>&1; echo "repro $i rc=$?" >> $O/summary.txt; done
timeout 300 python -m pytest tests/test_tas_cycle_engine.py tests/test_tas_closed_loop.py -m gpu -x -q -p no:cacheprovider > $O/behind.txt 2>&1; echo "behind rc=$?" >> $O/summary.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "gpu suite (one process) rc=$?" >> $O/summary.txt
tail -n 3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload cfg5-cycle --steps 20 --warmup 4 > $O/bench_cfg5_cycle.json 2> $O/bench_cfg5_cycle.err
timeout 300 python tools/prof_tas_closed.py > $O/prof_tas_closed.txt 2>&1; grep -h "sum of the entry\|kernel ms\|recomputation (get" $O/prof_tas_closed.txt
timeout 300 python tools/fuzz_tas_cycle.py 5000 5400 hip second > $O/fuzz_second_hip.txt 2>&1; tail -n 2 $O/fuzz_second_hip.txt
cat $O/summary.txt
cat $O/bench_default.json $O/bench_cfg5_cycle.json | cut -c1-330
