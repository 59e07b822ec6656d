cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
timeout 600 python bench.py --workload cfg5-cycle --steps 10 --warmup 2 --cpu-seconds 5 > gpurun_out/r03l/bench_cfg5cycle.json 2> gpurun_out/r03l/bench_cfg5cycle.err; tail -3 gpurun_out/r03l/bench_cfg5cycle.err; cut -c1-2500 gpurun_out/r03l/bench_cfg5cycle.json
