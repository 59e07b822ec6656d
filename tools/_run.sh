cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
(timeout 900 python -m pytest tests/test_tas_cycle_engine.py -m gpu -x -q > gpurun_out/r03l/gpu_tas_cycle.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03l/gpu_tas_cycle.log)
tail -15 gpurun_out/r03l/gpu_tas_cycle.log
