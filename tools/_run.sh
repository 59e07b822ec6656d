cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03o
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03o/gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03o/gpu_tests.log)
tail -4 gpurun_out/r03o/gpu_tests.log
python - <<PY
import sys; sys.path.insert(0,'.')
import bench, time
from kueue_amd.api import make_config
from kueue_amd.engine import Engine
from kueue_amd.population import generate
for cfgn in (3,4):
    pop=generate(cfgn); eng=Engine(make_config()); 
    ts=[]
    for i in range(5):
        t=time.perf_counter(); eng.put(pop.snapshot); ts.append((time.perf_counter()-t)*1e3)
    print("cfg",cfgn,"kq_snapshot_put ms",[round(x,2) for x in ts])
PY
