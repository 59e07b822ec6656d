cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pending_step.py tests/test_pending.py -m gpu -x -q > gpurun_out/r03j_gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03j_gpu_tests.log
tail -4 gpurun_out/r03j_gpu_tests.log
python - <<PY
import sys; sys.path.insert(0,'.')
import bench
from kueue_amd.api import make_config
from kueue_amd.engine import Engine
from kueue_amd.population import generate
pop=generate(3); eng=Engine(make_config()); eng.put(pop.snapshot)
print(bench.pending_cost(eng,pop))
PY
