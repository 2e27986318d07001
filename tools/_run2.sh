cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_specttuner.py -x -q 2>&1 | tail -5
ST_WAVE=1 timeout 300 python tools/st_bench.py 2>&1 | tail -8
