cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03n
(timeout 900 python -m pytest tests/test_tas_cycle_engine.py -m gpu -x -q > gpurun_out/r03n/gpu_tas.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03n/gpu_tas.log)
tail -3 gpurun_out/r03n/gpu_tas.log
timeout 600 python bench.py --workload cfg5-cycle --steps 10 --warmup 2 --cpu-seconds 5 > gpurun_out/r03n/bench_cfg5cycle.json 2> gpurun_out/r03n/bench_cfg5cycle.err; tail -2 gpurun_out/r03n/bench_cfg5cycle.err; python -c "
import json; d=json.load(open('gpurun_out/r03n/bench_cfg5cycle.json')); print(d['value'], d['kernel_ms_per_cycle'], d['parity_checked'], d['per_cycle'])"
