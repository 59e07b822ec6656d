#!/bin/bash
# round 5, session d: in-kernel segment timers (libkq_engine_prof.so, -DKQ_PROF) of the fair victim search after the path / LCA tables and the
# one-trip row fetch, process kernel only and with the nominate pass
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${KQ_TAG:-r05d}; mkdir -p $O
KQ_PROF_SKIP_NOMINATE=1 timeout 400 python tools/prof_fair.py 1000 > $O/prof_fair_cfg4f_process_only.txt 2>&1; grep "search:\|lds search\|recompute\|wall" $O/prof_fair_cfg4f_process_only.txt | grep -v " 0 cycles"
KQ_PROF_SKIP_NOMINATE=1 timeout 300 python tools/prof_cfg4c.py > $O/prof_cfg4c_process_only.txt 2>&1; grep -v " 0 cycles" $O/prof_cfg4c_process_only.txt | head -40
