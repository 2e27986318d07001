"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs: python tools/pmc_table.py <dir> [<dir> ...] [--match substr]"""
import csv, glob, os, sys
from collections import defaultdict
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    dirs = [d for d in dirs if d != match]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if match and match not in k:
                continue
            k = k.replace("void (anonymous namespace)::", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = sorted(acc)
names = sorted({c for k in kern for c in acc[k]})
print("| counter | " + " | ".join(kern) + " |")
print("|---|" + "---|" * len(kern))
for c in names:
    print(f"| {c} | " + " | ".join(("%.4g" % (sum(acc[k][c]) / len(acc[k][c]))) if acc[k][c] else "" for k in kern) + " |")
print("| launches | " + " | ".join(str(max((len(v) for v in acc[k].values()), default=0)) for k in kern) + " |")
