# chan_pair_kernel tile shapes: alone (tools/fir_c1.py) and behind the pipeline's other streams (bench.py --workload c2 --channeliser fir)
# usage: bash tools/nwsweep.sh [log2 of the bench's block, default 24]
B=${1:-24}
export FIR_LOG2L=22,23,24
for cfg in "auto 0" "8 0" "2 1" "4 1" "4 2"; do
  set -- $cfg
  if [ "$1" = "auto" ]; then unset SUAMD_FIR_PAIR_NW; else export SUAMD_FIR_PAIR_NW=$1; fi
  if [ "$2" = "0" ]; then unset SUAMD_FIR_PAIR_TPW; else export SUAMD_FIR_PAIR_TPW=$2; fi
  echo "== NW=$1 TPW=$2"
  python tools/fir_c1.py 2>/dev/null
  python bench.py --workload c2 --channeliser fir --block $B --steps 12 --warmup 3 --lean 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('   pipeline, block 2^$B:', r.get('kernel_ms'), 'frac', r.get('frac'), {k:(v['avg'],v['min'],v['max']) for k,v in r.get('kernel_launches_ms',{}).items() if 'chan' in k})"
done
