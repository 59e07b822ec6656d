#!/bin/bash
# round 5, session a: the whole GPU suite on the new build (fair search: LCA table + one-trip row fetch; kq_group N=2 on one device; guards),
# the group bench, the fair / classical preemption benches, the TAS closed loop, and the guard fuzz of kq_snapshot_put.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/box.txt; nproc >> $O/box.txt
timeout 900 python -m pytest tests -m gpu -x -q -n 2 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
KQ_GUARD=1 timeout 400 python tools/fuzz_put_guard.py --iters 10000 --seconds 300 --seed 1 > $O/fuzz_put_guard_seed1.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz_put_guard_seed1.txt
tail -4 $O/fuzz_put_guard_seed1.txt
KQ_GUARD=1 KQ_ROWS_TRACE=1 timeout 200 python tools/fuzz_put_guard.py --iters 3000 --seconds 90 --seed 2 > $O/fuzz_put_guard_trace_seed2.out 2> $O/fuzz_put_guard_trace_seed2.err; echo "fuzz-trace rc=$?" >> $O/fuzz_put_guard_trace_seed2.out
tail -3 $O/fuzz_put_guard_trace_seed2.out; wc -l $O/fuzz_put_guard_trace_seed2.err; tail -2 $O/fuzz_put_guard_trace_seed2.err > $O/fuzz_trace_last_lines.txt; grep -c "/ no error" $O/fuzz_put_guard_trace_seed2.err >> $O/fuzz_trace_last_lines.txt; grep -v "/ no error" $O/fuzz_put_guard_trace_seed2.err | head -20 >> $O/fuzz_trace_last_lines.txt; rm -f $O/fuzz_put_guard_trace_seed2.err
for w in cfg3-group cfg4c-group cfg4f-group; do timeout 300 python bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err; echo "$w rc=$?"; done
timeout 300 python bench.py --workload cfg4f --steps 3 --warmup 1 > $O/bench_cfg4f.json 2> $O/bench_cfg4f.err; echo "cfg4f rc=$?"
timeout 200 python bench.py --workload cfg4c --steps 10 --warmup 2 > $O/bench_cfg4c.json 2> $O/bench_cfg4c.err; echo "cfg4c rc=$?"
timeout 300 python bench.py --workload cfg5-cycle --steps 20 --warmup 4 > $O/bench_cfg5cycle_closed.json 2> $O/bench_cfg5cycle_closed.err; echo "cfg5-cycle rc=$?"
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout 200 python bench.py --workload cfg5-split --steps 8 --warmup 2 > $O/bench_cfg5split.json 2> $O/bench_cfg5split.err; echo "cfg5-split rc=$?"
python tools/codeobj_meta.py > $O/codeobj_metadata.txt 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp
Q="--no-cpu-baseline --no-parity-gate --full-run 0 --no-host-leg"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p_cfg4f -- python $R/bench.py --workload cfg4f --steps 1 --warmup 0 $Q > $R/$O/p_cfg4f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p_cfg3 -- python $R/bench.py $Q > $R/$O/p_cfg3.log 2>&1
cd $R
for w in cfg4f cfg3; do f=$(find $O/p_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${w}_kernel_stats.csv; rm -rf $O/p_$w; done
for f in $O/bench_*.json; do echo "== $f"; head -c 600 $f; echo; done
