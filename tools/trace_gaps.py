"""From a rocprofv3 --kernel-trace CSV: per kernel family, the gaps between consecutive launches (end -> next start), and
what the recurrence streams' pauses around the transform kernels look like.   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
import collections

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
t0 = rows[0][0]
by = collections.defaultdict(list)
for a, b, n in rows:
    by[n].append((a, b))
for n in ("costas_kernel", "clock_kernel", "agc_level_kernel", "stp_kernel", "psd_kernel"):
    v = by.get(n, [])
    if len(v) < 3:
        continue
    dur = [(b - a) / 1e3 for a, b in v]
    gap = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
    gap_s = sorted(gap)
    print(f"{n:18s} {len(v):4d} launches  dur avg {sum(dur) / len(dur):9.1f} us   gap to next launch: median {gap_s[len(gap_s) // 2]:8.1f}  "
          f"p90 {gap_s[int(len(gap_s) * 0.9)]:8.1f}  max {gap_s[-1]:8.1f} us   sum of gaps {sum(gap) / 1e3:.2f} ms over {(v[-1][1] - v[0][0]) / 1e6:.1f} ms")
# the pause of the Costas stream around each stp launch
cos = by.get("costas_kernel", [])
for a, b in by.get("stp_kernel", [])[5:15]:
    before = max((e for s, e in cos if e <= a), default=None)
    after = min((s for s, e in cos if s >= b), default=None)
    running = [1 for s, e in cos if s < a < e]
    psd = max((s for s, e in by.get("psd_kernel", []) if e <= a), default=a)
    if before and after:
        print(f"  stp at {(a - t0) / 1e6:8.3f} ms dur {(b - a) / 1e3:6.1f}: costas ended {(a - before) / 1e3:7.1f} us before (psd started {(a - psd) / 1e3:6.1f} before), "
              f"next costas {(after - b) / 1e3:7.1f} us after; costas running beside it: {len(running)}")
