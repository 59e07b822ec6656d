cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pending_step.py tests/test_pending.py tests/test_afs_ledger.py tests/test_sharded_cycle_gloo.py -m gpu -x -q > gpurun_out/r03g_gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03g_gpu_tests.log
tail -4 gpurun_out/r03g_gpu_tests.log
for loop in sync pipelined; do
python bench.py --workload cfg3 --steps 200 --warmup 10 --no-cpu-baseline --full-run 0 --no-host-leg --loop $loop 2>gpurun_out/r03g_$loop.err | grep '^{"metric' > gpurun_out/r03g_bench_cfg3_$loop.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03g_bench_cfg3_$loop.json"))
print("$loop", d["value"], d["ms_per_step"], d["p50_cycle_ms"], d["p99_cycle_ms"], d["kernel_ms_per_cycle"], d["parity_checked"], d["config"]["heads_per_cycle"])
PY
done
KQ_STEP_UNFUSED=1 python bench.py --workload cfg3 --steps 200 --warmup 10 --no-cpu-baseline --full-run 0 --no-host-leg --no-parity-gate 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unfused tail', d['value'], d['ms_per_step'])"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03g_cfg3 -o cfg3 -- python bench.py --workload cfg3 --steps 60 --warmup 10 --no-cpu-baseline --full-run 0 --no-parity-gate --no-host-leg > gpurun_out/r03g_cfg3.log 2>&1
python tools/cycle_timeline.py gpurun_out/r03g_cfg3 > gpurun_out/r03g_cfg3_timeline.txt 2>&1
cat gpurun_out/r03g_cfg3_timeline.txt
