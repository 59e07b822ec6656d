cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03p
(timeout 900 python -m pytest tests/test_tas_cycle_engine.py -m gpu -x -q > gpurun_out/r03p/gpu_tas.log 2>&1; echo "pytest exit $?" >> gpurun_out/r03p/gpu_tas.log)
tail -3 gpurun_out/r03p/gpu_tas.log
