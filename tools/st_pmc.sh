# SQ counter passes over the FFT channeliser kernel (C = D = 64, 4 Mi-sample block, C4 raster): bash tools/st_pmc.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/st_pmc
export ST_ONE=1 ST_SPACING=${ST_SPACING:-0.0036}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/st_pmc -o p -- python tools/st_bench.py > gpurun_out/st_pmc/out.txt 2> gpurun_out/st_pmc/err.txt
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/st_pmc2 -o p -- python tools/st_bench.py > gpurun_out/st_pmc/out2.txt 2> gpurun_out/st_pmc/err2.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/st_pmc3 -o p -- python tools/st_bench.py > gpurun_out/st_pmc/out3.txt 2> gpurun_out/st_pmc/err3.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/st_pmc4 -o p -- python tools/st_bench.py > gpurun_out/st_pmc/out4.txt 2> gpurun_out/st_pmc/err4.txt
python - <<'PY'
import csv, glob, collections
for d in ("st_pmc", "st_pmc2", "st_pmc3", "st_pmc4"):
    acc = collections.defaultdict(list)
    for fn in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if any(k in r["Kernel_Name"] for k in ("st_kernel", "stw_kernel", "stp_kernel")):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:28s} launches {len(v):3d}  mean {sum(v)/len(v):16.1f}")
PY
tail -2 gpurun_out/st_pmc/out.txt
