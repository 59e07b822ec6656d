#!/bin/bash
# Measurement pass of a round on the GPU box: benches (JSON lines) + rocprofv3 kernel stats + PMC passes (FETCH_SIZE / WRITE_SIZE in
# separate runs, never combined with other trace domains). Everything lands under gpurun_out/$1/.
TAG=${1:-r02g}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --workload cfg3-batch --steps 20 --warmup 2 > $O/bench_cfg3b.json 2> $O/bench_cfg3b.err
python bench.py --workload cfg3-split --steps 50 > $O/bench_cfg3split.json 2> $O/bench_cfg3split.err
python bench.py --workload cfg3-split --steps 50 --fill 0.3 > $O/bench_cfg3split_fill03.json 2> $O/bench_cfg3split_fill03.err
python bench.py --workload cfg3f --steps 30 --full-run 0 --no-host-leg > $O/bench_cfg3f.json 2> $O/bench_cfg3f.err
python bench.py --workload cfg4c --steps 3 --warmup 1 --cpu-seconds 5 > $O/bench_cfg4c.json 2> $O/bench_cfg4c.err
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-parity-gate --full-run 0 --no-host-leg"
for W in cfg3 cfg3-batch; do
  if [ $W = cfg3 ]; then CMD="python $R/bench.py $Q"; else CMD="python $R/bench.py --workload cfg3-batch --steps 10 --warmup 1 --no-cpu-baseline"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_${W}_stats -- $CMD > $O/p_${W}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_${W}_fetch -- $CMD > $O/p_${W}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_${W}_write -- $CMD > $O/p_${W}_write.log 2>&1
done
cd $R
if [ "$2" = "cfg4f" ]; then timeout 400 python bench.py --workload cfg4f --steps 1 --warmup 0 --no-cpu-baseline --no-host-leg > $O/bench_cfg4f.json 2> $O/bench_cfg4f.err; fi
for f in $O/bench_*.json; do echo "== $f"; cut -c1-900 $f; done
tail -n 2 $O/*.err | grep -v amdgpu.ids
