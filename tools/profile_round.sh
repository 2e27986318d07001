#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel-trace stats of the bench, then separate PMC passes
# (FETCH_SIZE / WRITE_SIZE -- never combined with other trace domains), for the default workload and C5.
# Usage (from the repo root, on the GPU box):  bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { ( cd $REPO && "$@" ); }
cd $REPO
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --lean > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_all -o t -- python bench.py --extra --no-pmc --no-cpu-baseline > $OUT/bench_all_under_rocprof.json 2> $OUT/trace_all.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --steps 4 --warmup 1 --lean > /dev/null 2> $OUT/pmc_$c.err
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmcall_$c -o p -- python bench.py --steps 4 --warmup 1 --extra --no-pmc --no-cpu-baseline > /dev/null 2> $OUT/pmcall_$c.err
done
# the driver's exact command (its compact line + the detail file), the default command (150 steps) with the secondary workloads
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench.err
cp bench_detail.json $OUT/bench_detail.json
python bench.py --extra > $OUT/bench.json 2>> $OUT/bench.err
cp bench_detail.json $OUT/bench_detail_extra.json
python bench.py --lean --isolated > $OUT/bench_isolated.json 2>> $OUT/bench.err
# the bank clock recovery, aligned / staggered symbol clocks, in its three schedules
for m in 2 1 0; do SUAMD_CLOCK_MODE=$m python tools/clock_stagger.py 2>/dev/null; done > $OUT/clock_staggered.txt
# the boundary: BASELINE configs[3]'s per-GPU slice through the suscan ABI, and rocprofv3's view of its kernels
for lg in 21 22; do LIVE_LOG2=$lg python tools/live_c4.py 2>/dev/null | tail -1; done > $OUT/live_analyzer.txt
for s in 0; do SUAMD_ST_ROW_STAGE=$s LIVE_LOG2=21 python tools/live_c4.py 2>/dev/null | tail -1; done >> $OUT/live_analyzer.txt
(cd /tmp && LIVE_LOG2=21 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_live -o t -- python $REPO/tools/live_c4.py > /dev/null 2> $OUT/trace_live.err)
f=$(find $OUT/trace_live -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/live_kernel_stats.csv
# the same leg with the inspectors as slab columns (default) and as rows (SUAMD_ANALYZER_SLAB=0), 2 Mi and 4 Mi blocks: rate + rocprofv3's kernels
: > $OUT/live_slab_ab.txt
for v in 1 0; do for lg in 21 22; do
  echo "SUAMD_ANALYZER_SLAB=$v block 2^$lg:" >> $OUT/live_slab_ab.txt
  SUAMD_ANALYZER_SLAB=$v LIVE_LOG2=$lg python tools/live_c4.py 2>/dev/null | tail -1 >> $OUT/live_slab_ab.txt
  (cd /tmp && SUAMD_ANALYZER_SLAB=$v LIVE_LOG2=$lg LIVE_BLOCKS=16 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_live_${v}_$lg -o t -- python $REPO/tools/live_c4.py > /dev/null 2>> $OUT/trace_live.err)
  f=$(find $OUT/trace_live_${v}_$lg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -v "at::native\|rocclr" "$f" | head -30 > profiles/${TAG}_live_slab${v}_block${lg}_kernel_stats.csv
done; done
cp $OUT/live_slab_ab.txt profiles/${TAG}_live_slab_ab.txt
python tools/gang_bench.py > profiles/${TAG}_gang_bench.txt 2>/dev/null
python tools/st_bench.py > $OUT/kernel_microbench.txt 2>&1
python tools/fir_bench.py >> $OUT/kernel_microbench.txt 2>&1
python tools/fir_c1.py >> $OUT/kernel_microbench.txt 2>&1
SUAMD_FIR_STREAM=0 python tools/fir_c1.py >> $OUT/kernel_microbench.txt 2>&1
python tools/st_wide.py >> $OUT/kernel_microbench.txt 2>&1
python tools/psd_bench.py >> $OUT/kernel_microbench.txt 2>&1
bash tools/st_pmc.sh > $OUT/st_sq_counters.txt 2>&1
python tools/psd_large.py > $OUT/psd_large_frames.txt 2>&1
SUAMD_PSD_LARGE=passes python tools/psd_large.py > $OUT/psd_large_frames_passes.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmcl_$c -o p -- python tools/psd_large.py > /dev/null 2> $OUT/pmcl_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_psdl -o t -- python tools/psd_large.py > /dev/null 2> $OUT/trace_psdl.err
python tools/prof_summary.py $OUT $TAG
cp $OUT/st_sq_counters.txt profiles/${TAG}_st_sq_counters.txt 2>/dev/null
python - <<PY
import csv, glob, collections, json
out = "$OUT"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob(f"{out}/pmcl_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == c and ("psdl_" in r["Kernel_Name"] or "psd_kernel" in r["Kernel_Name"]):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] + f" [grid {r['Grid_Size']}]"
                acc[k][c].append(float(r["Counter_Value"]))
res = {k: {"launches": len(v["FETCH_SIZE"]), "FETCH_SIZE_KiB": sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])),
           "WRITE_SIZE_KiB": sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])),
           "hbm_bytes_per_launch": int(2048 * sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) + 1024 * sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])))}
       for k, v in sorted(acc.items())}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/psd_large.py: 2^28 samples per feed, "
                     "frames of 16384 (in-LDS kernel) .. 1048576 points (psd_large.hip); per-launch averages, KiB; "
                     "HBM bytes = 2 FETCH + WRITE (gfx950 half-count of streamed reads)", "kernels_by_grid": res},
          open(f"profiles/${TAG}_psd_large_pmc.json", "w"), indent=1)
PY
cp $OUT/psd_large_frames.txt profiles/${TAG}_psd_large_frames.txt
echo "--- round 2's path (SUAMD_PSD_LARGE=passes: radix-16 passes through HBM) ---" >> profiles/${TAG}_psd_large_frames.txt
cat $OUT/psd_large_frames_passes.txt >> profiles/${TAG}_psd_large_frames.txt
f=$(find $OUT/trace_psdl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" profiles/${TAG}_psd_large_kernel_stats.csv
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
ls -la $OUT
