import sys, json
d = json.loads(sys.stdin.read())
print(d["value"], d["roofline"]["stage_ms"])
for k, v in d["other_workloads"].items():
    if k == "c5":
        continue
    for name, e in (v.items() if "value_MSps" not in v else [("", v)]):
        r = e.get("roofline", {})
        print(k, name, e["value_MSps"], e.get("stage_ms"), "FIR-stage frac", r.get("frac"), "alone:", r.get("fir_stage_alone", {}).get("frac"), "16 Mi:", r.get("fir_stage_16Mi_block", {}).get("frac"))
