cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_specttuner.py tests/test_gpu_analyzer_fft.py tests/test_gpu_fullsize_oracle.py -x -q 2>&1 | tail -3
ST_ONE=1 python tools/st_bench.py 2>&1 | tail -1
python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['stage_ms'])"
