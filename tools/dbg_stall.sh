#!/bin/bash
# which HIP runtime call takes the milliseconds of the default bench's step 48
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04q_stall; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --output-format csv -d $O/trace -- python $R/bench.py --steps 60 --no-cpu-baseline --full-run 0 --no-host-leg --no-parity-gate > $O/log.txt 2>&1
f=$(ls $O/trace/*/*hip_api_trace.csv | head -1)
python - <<P
import csv
rows=list(csv.DictReader(open("$f")))
print(len(rows), rows[0].keys())
slow=[r for r in rows if int(r["End_Timestamp"])-int(r["Start_Timestamp"])>1_000_000]
t0=int(rows[0]["Start_Timestamp"])
for r in slow[-25:]:
    print(r["Function"], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, "ms at", (int(r["Start_Timestamp"])-t0)/1e9, "s")
P
cp $f $R/gpurun_out/r04q_stall_hip_api_trace.csv 2>/dev/null; gzip -f $R/gpurun_out/r04q_stall_hip_api_trace.csv
