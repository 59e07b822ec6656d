#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05zz; mkdir -p $O
timeout 90 python -m pytest tests/test_tas_cycle_engine.py -m gpu -x -q -p no:cacheprovider -k "schedule_tas_gpu and partial" > $O/pytest_partial_cases.txt 2>&1; echo "rc=$?" >> $O/pytest_partial_cases.txt; tail -n 3 $O/pytest_partial_cases.txt
