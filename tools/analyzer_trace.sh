#!/bin/bash
# Kernel timeline of the live analyzer (tools/analyzer_bench.py) for the stage-overlap analysis in DESIGN.md.
# Usage (on the GPU box): bash tools/analyzer_trace.sh [N inspectors]; writes gpurun_out/antrace/trace_small.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-64}
export PYTHONPATH=$R
mkdir -p $R/gpurun_out/antrace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/antrace -- python $R/tools/analyzer_bench.py $N 12 2>&1 | grep inspectors
cd $R/gpurun_out/antrace
for f in $(find . -name "*kernel_trace.csv"); do
python - "$f" <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), list(rows[0].keys()))
keep = [(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70].replace(",", ";"), r.get("Stream_Id", r.get("Queue_Id")), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
keep.sort(key=lambda k: k[2])
t0 = keep[0][2]
with open("trace_small.csv", "w") as f:
    for k in keep[-8000:]:
        f.write("%s,%s,%d,%d\n" % (k[0], k[1], k[2] - t0, k[3] - t0))
PY
rm -f $f
done
find . -name "*.csv" ! -name trace_small.csv -delete
