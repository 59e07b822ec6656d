#!/bin/bash
# round 5, final session: the committed build — the whole GPU suite, every bench workload (tools/prof_round.sh benches), rocprofv3 kernel stats
# + PMC passes of the default line, cfg 4f and the closed TAS loop, the code-object metadata
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -n 2 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -n 3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
bash tools/prof_round.sh r05f benches none 2>&1 | cut -c1-400 | tail -n 60
PROF_WORKLOADS="cfg3 cfg4f cfg5-cycle" bash tools/prof_round.sh r05f none profiles 2>&1 | tail -n 6
python tools/codeobj_meta.py > $O/codeobj_metadata.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
