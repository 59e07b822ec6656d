#!/bin/bash
# One GPU session of round 6: usage  tools/gpu_session.sh <tag> <step> [<step> ...]
#   steps: tests (the whole -m gpu suite), smoke, bench (default bench line), bench:<workload>[:extra args], py:<script and args>
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  case "$step" in
    tests) timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "gpu suite rc=$?" >> $O/summary.txt; tail -n 3 $O/pytest_gpu.txt ;;
    tests:*) K="${step#tests:}"; timeout 1200 python -m pytest tests -x -q -m gpu -k "$K" > "$O/pytest_gpu_k.txt" 2>&1; echo "gpu suite -k '$K' rc=$?" >> $O/summary.txt; tail -n 3 $O/pytest_gpu_k.txt ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/summary.txt; tail -n 1 $O/smoke.txt ;;
    bench) timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" >> $O/summary.txt; tail -c 1500 $O/bench_default.json ;;
    bench:*) IFS=: read -r _ W X <<< "$step"; timeout 2400 python bench.py --workload $W $X > $O/bench_$W.json 2> $O/bench_$W.err; echo "bench $W $X rc=$?" >> $O/summary.txt; tail -c 1500 $O/bench_$W.json ;;
    py:*) S="${step#py:}"; N=$(echo "$S" | tr ' /' '__' | cut -c1-60); timeout 2400 python $S > $O/py_$N.txt 2>&1; echo "python $S rc=$?" >> $O/summary.txt; tail -n 15 $O/py_$N.txt ;;
  esac
done
cat $O/summary.txt
