# the headline workload at 4 / 8 / 16 Mi-sample blocks, default slot plan and all 1024 slots: the channeliser's own duration in the pipeline
for b in 22 23 24; do for s in 0 896 1024; do
  if [ $s = 0 ]; then unset SUAMD_ST_SLOTS; else export SUAMD_ST_SLOTS=$s; fi
  python bench.py --block $b --steps $((2400 >> (b - 20))) --lean 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('block 2^$b slots $s:', j['value'], 'MS/s  kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), {k:(round(v['avg']*1e3,1),round(v['min']*1e3,1),round(v['max']*1e3,1)) for k,v in r.get('kernel_launches_ms',{}).items()}, 'psd frac', r['psd']['frac'])"
done; done
