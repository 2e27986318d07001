for v in "" "SUAMD_PSD_SPLIT_TARGET=768" "SUAMD_PSD_SPLIT_TARGET=1024" "SUAMD_PSD_SPLIT_TARGET=2048 SUAMD_PSD_MIN_FRAMES=1" "SUAMD_PSD_SPLIT_TARGET=256" "SUAMD_ST_SLOTS=704" "SUAMD_ST_SLOTS=640" "SUAMD_ST_SLOTS=512"; do
  echo "== $v"
  env $v python bench.py --steps 40 --warmup 5 --lean 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('  ', j['value'], 'stp frac', r.get('frac'), 'psd frac', r['psd']['frac'], {k:(round(v['avg']*1e3,1),round(v['min']*1e3,1),round(v['max']*1e3,1)) for k,v in r.get('kernel_launches_ms',{}).items()})"
done
