R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02w
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload cfg4f --steps 1 --warmup 0 --no-cpu-baseline --no-host-leg --no-parity-gate"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_cfg4f_stats -- $CMD > $O/p_cfg4f_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_cfg4f_fetch -- $CMD > $O/p_cfg4f_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_cfg4f_write -- $CMD > $O/p_cfg4f_write.log 2>&1
ls $O; tail -n 2 $O/p_cfg4f_stats.log
