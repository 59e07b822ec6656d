#!/bin/bash
# r05l: the build with second-pass heads in the TAS cycle: full GPU suite, smoke, the default line, the TAS cycle lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05l; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -n 2 > $O/pytest_gpu.txt 2>&1
tail -n 4 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload cfg5-cycle --steps 20 --warmup 4 > $O/bench_cfg5_cycle.json 2> $O/bench_cfg5_cycle.err
timeout 300 python bench.py --workload cfg5f-cycle --steps 10 --warmup 2 > $O/bench_cfg5f_cycle.json 2> $O/bench_cfg5f_cycle.err
timeout 200 python tools/fuzz_tas_cycle.py 5000 5400 hip second > $O/fuzz_second_hip.txt 2>&1; tail -n 2 $O/fuzz_second_hip.txt
cat $O/bench_default.json $O/bench_cfg5_cycle.json $O/bench_cfg5f_cycle.json | cut -c1-330
