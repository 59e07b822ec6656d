#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05zz; mkdir -p $O
timeout 70 python -m pytest tests/test_oracle_tas.py tests/test_tas_replacement.py -m gpu -x -q -p no:cacheprovider -k "taint or (exclusion and stats) or matches" > $O/pytest_taint_cases.txt 2>&1; echo "rc=$?" >> $O/pytest_taint_cases.txt; tail -n 3 $O/pytest_taint_cases.txt
