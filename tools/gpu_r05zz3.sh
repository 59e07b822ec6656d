#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05zz; mkdir -p $O
timeout 80 python -m pytest tests/test_oracle_tas.py -m gpu -x -q -p no:cacheprovider -k "engine_gpu and (selector or leakage)" > $O/pytest_node_selector_cases.txt 2>&1; echo "rc=$?" >> $O/pytest_node_selector_cases.txt; tail -n 3 $O/pytest_node_selector_cases.txt
