cd $GRAFT_REPO_ROOT
SUAMD_STW_TSTAMP=1 ST_ONE=1 timeout 300 python tools/st_bench.py 2>&1 | grep -A3 "stw tstamp" | tail -8
SUAMD_STW_TSTAMP=1 ST_ONE=1 ST_RUN=4 timeout 300 python tools/st_bench.py 2>&1 | grep -A5 "stw tstamp" | tail -6
