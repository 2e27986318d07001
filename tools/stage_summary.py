"""One line per workload of a bench.py JSON line (stdin): rate, stage times, FIR-stage fraction in the pipeline and alone."""
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
alone = lambda rr: {k: (v.get("frac") if isinstance(v, dict) else v) for k, v in (rr.get("fir_stage_alone") or {}).items()}
print("main", d["config"].get("block_samples"), d["value"], r["stage_ms"], "FIR-stage frac", r.get("frac"), "alone:", alone(r),
      "psd frac", r.get("psd_kernel", {}).get("frac"), "traffic", r.get("traffic"))
for k, v in d.get("other_workloads", {}).items():
    if k == "c5":
        print(k, v.get("value_MSps"), v.get("psd_roofline", {}).get("frac"))
        continue
    for name, e in (v.items() if "value_MSps" not in v else [("", v)]):
        rr = e.get("roofline", {})
        print(k, name, e.get("value_MSps"), e.get("stage_ms"), "FIR-stage frac", rr.get("frac"), "alone:", alone(rr))
