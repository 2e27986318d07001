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
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --no-extra --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_all -o t -- python bench.py --no-cpu-baseline > $OUT/bench_all_under_rocprof.json 2> $OUT/trace_all.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python bench.py --steps 4 --warmup 1 --no-extra --no-cpu-baseline > /dev/null 2> $OUT/pmc_$c.err
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmcall_$c -o p -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmcall_$c.err
done
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --no-extra --no-cpu-baseline --isolated > $OUT/bench_isolated.json 2>> $OUT/bench.err
python tools/st_bench.py > $OUT/kernel_microbench.txt 2>&1
python tools/fir_bench.py >> $OUT/kernel_microbench.txt 2>&1
python tools/psd_bench.py >> $OUT/kernel_microbench.txt 2>&1
bash tools/st_pmc.sh > $OUT/st_sq_counters.txt 2>&1
python tools/prof_summary.py $OUT $TAG
cp $OUT/st_sq_counters.txt profiles/${TAG}_st_sq_counters.txt 2>/dev/null
ls -la $OUT
