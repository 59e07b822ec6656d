#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05j; mkdir -p $O
timeout 600 python tools/prof_tas_closed.py > $O/prof_tas_closed.txt 2>&1; cat $O/prof_tas_closed.txt | tail -n 45
