#!/bin/bash
# round 4, session b: the batched fair search / lazy classical search on the MI355X: GPU suite, then the preemption benches
O=gpurun_out/r04b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; cut -c1-900 $O/bench_$name.json; tail -2 $O/bench_$name.err | grep -v amdgpu.ids; }
run cfg4c --workload cfg4c --steps 3 --warmup 1 --cpu-seconds 5
TMO=900 run cfg4f --workload cfg4f --steps 1 --warmup 0 --cpu-seconds 5 --no-host-leg
run cfg3f --workload cfg3f --steps 30 --full-run 0 --no-host-leg
