cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pending_step.py tests/test_pending.py tests/test_afs_ledger.py -m gpu -x -q > gpurun_out/r03h_gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03h_gpu_tests.log
tail -4 gpurun_out/r03h_gpu_tests.log
python bench.py --workload cfg3 --steps 200 --warmup 10 --no-cpu-baseline --full-run 0 --no-host-leg --parity-cycles 200 2>gpurun_out/r03h.err | grep '^{"metric' > gpurun_out/r03h_bench_cfg3.json
python - <<PY
import json
d=json.load(open("gpurun_out/r03h_bench_cfg3.json"))
print("pipelined", d["value"], d["ms_per_step"], d["p50_cycle_ms"], d["p99_cycle_ms"], d["kernel_ms_per_cycle"], d["parity_checked"], d["parity"][:60], d["config"]["heads_per_cycle"])
PY
KQ_STEP_ONE_STREAM=1 python bench.py --workload cfg3 --steps 200 --warmup 10 --no-cpu-baseline --full-run 0 --no-host-leg --no-parity-gate 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one stream', d['value'], d['ms_per_step'])"
python bench.py --workload cfg2 --steps 200 --warmup 10 --no-cpu-baseline --full-run 0 --no-host-leg 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'], d['parity_checked'])"
python bench.py --workload cfg3f --steps 20 --warmup 2 --no-cpu-baseline --full-run 0 --no-host-leg 2>/dev/null | grep '^{"metric' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3f', d['value'], d['ms_per_step'], d['parity_checked'])"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03h_cfg3 -o cfg3 -- python bench.py --workload cfg3 --steps 60 --warmup 10 --no-cpu-baseline --full-run 0 --no-parity-gate --no-host-leg > gpurun_out/r03h_cfg3.log 2>&1
python tools/cycle_timeline.py gpurun_out/r03h_cfg3 > gpurun_out/r03h_cfg3_timeline.txt 2>&1
cat gpurun_out/r03h_cfg3_timeline.txt
