mkdir -p gpurun_out/r04a
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/gpu_tests.log 2>&1; echo "gpu tests rc=$?" 
tail -3 gpurun_out/r04a/gpu_tests.log
timeout 300 python tools/prof_cfg4c.py 1000 2 > gpurun_out/r04a/prof_cfg4c.txt 2>&1; tail -45 gpurun_out/r04a/prof_cfg4c.txt
timeout 400 python tools/prof_fair.py 1000 > gpurun_out/r04a/prof_fair_cfg4f.txt 2>&1; tail -50 gpurun_out/r04a/prof_fair_cfg4f.txt
