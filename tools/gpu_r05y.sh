#!/bin/bash
# r05y: the committed tree's final build once more: the GPU suite in one process as the driver runs it, smoke, the default line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05y; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "gpu suite (one process) rc=$?" >> $O/summary.txt; tail -n 2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/summary.txt; tail -n 1 $O/smoke.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" >> $O/summary.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_driver.json 2> $O/bench_default_driver.err
cat $O/summary.txt; cut -c1-330 $O/bench_default.json; cut -c1-330 $O/bench_default_driver.json
