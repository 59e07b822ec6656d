#!/bin/bash
# round 4, session c: where does the fair cycle's time go after the batching? A/B switches + in-kernel segment timers
O=gpurun_out/r04c; mkdir -p $O
run() { name=$1; shift; timeout ${TMO:-600} python bench.py "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "== $name rc=$?"; python - <<PY
import json
d=json.load(open("$O/bench_$name.json"))
print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_cycle","parity_checked")})
PY
}
Q="--no-cpu-baseline --full-run 0 --no-host-leg"
TMO=900 run cfg4f --workload cfg4f --steps 1 --warmup 0 $Q
KQ_FS_BATCH=0 TMO=900 run cfg4f_nobatch --workload cfg4f --steps 1 --warmup 0 $Q --no-parity-gate
run cfg3f --workload cfg3f --steps 30 $Q
KQ_FS_LRUN=0 run cfg3f_nolrun --workload cfg3f --steps 30 $Q --no-parity-gate
run cfg4c --workload cfg4c --steps 3 --warmup 1 $Q
timeout 400 python tools/prof_fair.py 1000 > $O/prof_fair_cfg4f.txt 2>&1; grep -v "0 cycles" $O/prof_fair_cfg4f.txt | head -60
timeout 300 python tools/prof_cfg4c.py 1000 2 > $O/prof_cfg4c.txt 2>&1; grep -v " 0 cycles" $O/prof_cfg4c.txt | head -45
