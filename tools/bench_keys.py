"""one line of the figures that matter out of a bench.py JSON line on stdin:  python bench.py ... | python tools/bench_keys.py tag"""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for ln in sys.stdin:
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    r = d.get("roofline", {})
    kl = r.get("kernel_launches_ms", {})
    parts = [f"{tag:28s} {d.get('value')} MS/s  step {d.get('ms_per_step')} ms  frac {r.get('frac')}"]
    for k, v in kl.items():
        parts.append(f"{k} avg {v.get('avg_all_samples', v['avg']) * 1e3:.1f} (min {v['min'] * 1e3:.1f} max {v['max'] * 1e3:.1f}) us x{v['launches']}")
    pk = r.get("psd", r.get("psd_kernel", {}))
    parts.append(f"psd frac {pk.get('frac')}")
    parts.append("stages " + json.dumps(d.get("stage_ms", r.get("stage_ms"))))
    print(" | ".join(parts), flush=True)
