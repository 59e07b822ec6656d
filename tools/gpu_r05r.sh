#!/bin/bash
# r05r: kq_tas_find_replacement with the required domain as a leaf range (no n x leaves mask): parity + timing at the cfg 5 topology, its GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05r; mkdir -p $O
timeout 300 python tools/bench_tas_replacement.py 2000 > $O/bench_tas_replacement.json 2> $O/bench_tas_replacement.err
timeout 300 python tools/bench_tas_replacement.py 2000 > $O/bench_tas_replacement_2.json 2>> $O/bench_tas_replacement.err
timeout 600 python -m pytest tests/test_tas_replacement.py tests/test_tas_elastic.py -m gpu -x -q -p no:cacheprovider > $O/pytest_repl.txt 2>&1; tail -n 2 $O/pytest_repl.txt
cat $O/bench_tas_replacement.json $O/bench_tas_replacement_2.json | cut -c1-900
