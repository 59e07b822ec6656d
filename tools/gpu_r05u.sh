#!/bin/bash
# r05u (kernel variants): balanced placement on the HIP engine (the reference's gated cases, random batches, random cycles), the benches that carry the
# placement (no regression from the new code path), then the whole GPU suite in one process + smoke on this build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05u; mkdir -p $O
timeout 900 python -m pytest tests/test_oracle_tas.py tests/test_emu_tas_random.py tests/test_tas_replacement.py -m gpu -x -q -p no:cacheprovider > $O/pytest_balanced_batch.txt 2>&1; echo "balanced batch rc=$?" >> $O/summary.txt; tail -n 2 $O/pytest_balanced_batch.txt
timeout 900 python -m pytest tests/test_tas_cycle_engine.py -m gpu -x -q -p no:cacheprovider -k "balanced" > $O/pytest_balanced_cycle.txt 2>&1; echo "balanced cycle rc=$?" >> $O/summary.txt; tail -n 2 $O/pytest_balanced_cycle.txt
timeout 300 python bench.py --workload cfg5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout 300 python bench.py --workload cfg5-cycle --steps 20 --warmup 4 > $O/bench_cfg5_cycle.json 2> $O/bench_cfg5_cycle.err
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "gpu suite (one process) rc=$?" >> $O/summary.txt; tail -n 2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
cat $O/summary.txt
cat $O/bench_cfg5.json $O/bench_cfg5_cycle.json $O/bench_default.json | cut -c1-260
