#!/bin/bash
# r05zz: the five delayed-topology second-pass cases (ProvisioningRequest) + every second-pass whole-cycle case on the HIP engine
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05zz; mkdir -p $O
timeout 100 python -m pytest tests/test_tas_cycle_engine.py -m gpu -x -q -p no:cacheprovider -k "schedule_tas_gpu and (ProvisioningRequest or second)" > $O/pytest_second_pass_cases.txt 2>&1; echo "rc=$?" >> $O/pytest_second_pass_cases.txt; tail -n 3 $O/pytest_second_pass_cases.txt
