set -u
O=gpurun_out/r6an; mkdir -p $O
export TMPDIR=/tmp; R=$(pwd)
for lg in 22 21; do for sl in 768 1024 896 640; do
  (cd /tmp && SUAMD_ST_SLOTS=$sl LIVE_LOG2=$lg LIVE_BLOCKS=16 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr -o t -- python $R/tools/live_c4.py > $R/$O/live.txt 2> $R/$O/live.err)
  f=$(find $O/tr -name "*kernel_stats.csv" | head -1)
  echo "== block 2^$lg slots $sl: $(tail -1 $O/live.txt | cut -c1-90)"
  python - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    m = re.search(r"(\w+_kernel)", n)
    if m and m.group(1) in ("stp_kernel", "psd_kernel", "costas_gang_slab_kernel"):
        print(f'   {m.group(1):28s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
PY
  rm -rf $O/tr
done; done 2>&1 | tee $O/slots.txt
