"""Condenses the rocprofv3 outputs of tools/profile_round.sh into profiles/<tag>_*.{md,json,csv}."""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


OURS = re.compile(r"^(st_kernel|stw_kernel|stp_kernel|chandet_|audio_|psd_|psdl_|chan_fir|chan_pair|costas_|clock_|agc_|pll_|quad_|xlate_|modulate_|update_hist|interpolate_|sweep_linear|"
                  r"feed_|fft_pass|frame_|window_pad|power_argmax|centroid|ingest|rows_|cma_|zc_|conj_prev|fac_|"
                  r"histogram_|delayed_|sample_manual|averager_|insp_spectrum|psd_shift)")
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
rows = []
if stats:
    shutil.copy(stats[0], os.path.join(prof, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        rows.append(r)
bench_line = ""
try:
    bench_line = open(os.path.join(out, "bench_under_rocprof.json")).read().strip().splitlines()[-1]
    bj = json.loads(bench_line)
except Exception:
    bj = None
block = bj["config"]["block_samples"] if bj else None
with open(os.path.join(prof, f"{tag}_kernel_stats_summary.md"), "w") as f:
    def table(rws):
        f.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
        for r in rws:
            n = short(r["Name"])
            if not OURS.match(n):
                continue
            f.write(f"| {n} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                    f"{float(r['MaxNs'])/1e3:.1f} | {r['Percentage']} |\n")
    f.write(f"# rocprofv3 --kernel-trace --stats, {tag} (MI355X)\n\n")
    f.write("`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --lean`\n"
            "(tools/profile_round.sh): the default workload (C4 slice: 8192-pt PSD + 64 QPSK inspectors; "
            f"{block} samples per block, {bj['steps'] if bj else '?'} steps + {bj['warmup'] if bj else '?'} warm-up).\n"
            f"Our kernels only; the full table (with torch's synthetic-data kernels) is `{tag}_kernel_stats.csv`.\n\n")
    if bj:
        f.write(f"Bench line under the profiler: value = {bj['value']} MS/s, stage_ms = {bj.get('stage_ms', bj['roofline'].get('stage_ms'))}; "
                f"the bench's own timer for the channeliser in the same run: {bj['roofline'].get('kernel_ms')} ms "
                f"(min / max {bj['roofline'].get('kernel_ms_min_max')}).\n\n")
    table(rows)
    f.write("\nThe serial (one-lane-per-channel) kernels run concurrently on separate streams, so their percentages add up "
            "to more than the wall time; the step time is the slowest of them.\n")
    stats_all = glob.glob(os.path.join(out, "trace_all", "**", "*kernel_stats.csv"), recursive=True)
    if stats_all:
        f.write("\n## with the secondary workloads (C2, C3, C5 and the live analyzer with 64 inspectors after the default one)\n\n"
                "`python bench.py --extra --no-pmc --no-cpu-baseline` under the same profiler; kernels shared by several workloads aggregate all of them\n"
                "(psd_kernel<13, 256, true> is C5's 8.6 GB launch, stp_kernel<6, true, .> includes the bench's launches of the\n"
                "kernel alone on 4 Mi and 16 Mi blocks, the recurrence kernels C2's 16x longer rows).\n\n")
        table(list(csv.DictReader(open(stats_all[0]))))

# PMC traffic: per kernel, average FETCH_SIZE / WRITE_SIZE (KiB) over launches
def collect(prefix, by_grid):
    pm = defaultdict(lambda: defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for fn in glob.glob(os.path.join(out, f"{prefix}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] != c:
                    continue
                k = short(r["Kernel_Name"])
                if not OURS.match(k):
                    continue
                k = k + f" [grid {r['Grid_Size']}]" if by_grid else k.split("<")[0]
                pm[k][c].append(float(r["Counter_Value"]))
    res = {}
    for k, d in sorted(pm.items()):
        fe = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1)
        wr = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
        res[k] = {"launches": len(d["FETCH_SIZE"]), "FETCH_SIZE_KiB": round(fe, 2), "WRITE_SIZE_KiB": round(wr, 2),
                  "hbm_bytes_per_launch": int(2 * fe * 1024 + wr * 1024)}
    return res


def _prev_calibration():
    for f in sorted(glob.glob(os.path.join(prof, "*_pmc_traffic.json")), reverse=True):
        try:
            c = json.load(open(f)).get("calibration")
        except Exception:
            c = None
        if c:
            return c
    return None


best = collect("pmc", False)          # default workload only (--no-extra): what bench.py's roofline.traffic reads
res = collect("pmcall", True)         # all workloads, split by launch grid (C2 / C3 / C5 launches differ in grid)
old = {}
p_old = os.path.join(prof, f"{tag}_pmc_traffic.json")
if os.path.exists(p_old):
    try:
        old = json.load(open(p_old))
    except Exception:
        old = {}
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 4 --warmup 1 "
                 "--no-cpu-baseline` (tools/profile_round.sh), MI355X",
       "units": "counter values are KiB; per-launch averages; HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
                "(FETCH_SIZE counts half of the streamed bytes on gfx950, see calibration)",
       "calibration": old.get("calibration") or _prev_calibration(),
       "workload": {"name": "c4", "block_samples": block},
       "kernels_by_grid": res}
doc["kernels"] = best
json.dump(doc, open(p_old, "w"), indent=1)
for a, b in (("bench.json", f"{tag}_bench.json"), ("bench_isolated.json", f"{tag}_bench_isolated.json"),
             ("bench_driver_command.json", f"{tag}_bench_driver_command.json"), ("bench_detail.json", f"{tag}_bench_detail.json"),
             ("bench_detail_extra.json", f"{tag}_bench_detail_extra.json"), ("clock_staggered.txt", f"{tag}_clock_staggered.txt"),
             ("live_analyzer.txt", f"{tag}_live_analyzer.txt"), ("live_kernel_stats.csv", f"{tag}_live_kernel_stats.csv"),
             ("bench_under_rocprof.json", f"{tag}_bench_under_rocprof.json"), ("kernel_microbench.txt", f"{tag}_kernel_microbench.txt")):
    src = os.path.join(out, a)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(prof, b))
print("profiles written for", tag)
