cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_analyzer.py -x -q 2>&1 | tail -3
python bench.py --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], d['roofline']['frac'], d['roofline']['stage_ms'])"
