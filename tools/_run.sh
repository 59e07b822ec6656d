cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03k
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03k/gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03k/gpu_tests.log)
tail -3 gpurun_out/r03k/gpu_tests.log
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name"; cut -c1-400 $O/bench_$name.json; }
TMO=900 run cfg3
run cfg2 --workload cfg2 --full-run 0
run cfg3f --workload cfg3f --steps 30 --full-run 0 --no-host-leg
run cfg3b --workload cfg3-batch --steps 20 --warmup 2
run cfg5 --workload cfg5 --steps 5 --warmup 1
PROF_WORKLOADS="cfg3" bash tools/prof_round.sh r03k none profiles
