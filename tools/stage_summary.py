import json,sys
d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["stage_ms"]); [print(k, v["value_MSps"], v.get("stage_ms")) for k,v in d["other_workloads"].items() if k!="c5"]
