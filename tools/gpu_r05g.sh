#!/bin/bash
# round 5, session g: segment timers with the per-row constants record (FsRowC) on and off (KQ_FS_ROWC_OFF), nextTarget cache still in this build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05g; mkdir -p $O
KQ_PROF_SKIP_NOMINATE=1 timeout 400 python tools/prof_fair.py 1000 > $O/prof_fair_cfg4f_rowc_on.txt 2>&1; grep "first strategy\|lds search\|fillBack\|wall" $O/prof_fair_cfg4f_rowc_on.txt | grep -v " 0 cycles"
KQ_FS_ROWC_OFF=1 KQ_PROF_SKIP_NOMINATE=1 timeout 400 python tools/prof_fair.py 1000 > $O/prof_fair_cfg4f_rowc_off.txt 2>&1; grep "first strategy\|lds search\|fillBack\|wall" $O/prof_fair_cfg4f_rowc_off.txt | grep -v " 0 cycles"
