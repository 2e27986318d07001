cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_specttuner.py tests/test_gpu_analyzer_fft.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5
ST_WAVE=1 python tools/st_bench.py 2>&1 | tail -8
ST_ONE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stw_trace -o t -- python tools/st_bench.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for fn in glob.glob("gpurun_out/stw_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "st" in r["Name"][:40] or "copy" in r["Name"].lower():
            print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
