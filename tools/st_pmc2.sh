# SQ counter passes over the channeliser kernels (pair / wave), C = D = 64, 4 Mi-sample block: bash tools/st_pmc2.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ST_ONE=1
for kern in pair wave; do
  export SUAMD_ST_KERNEL=$kern
  rm -rf gpurun_out/stp_pmc_$kern; mkdir -p gpurun_out/stp_pmc_$kern
  n=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
    n=$((n+1))
    timeout 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/stp_pmc_$kern/p$n -o p -- python tools/st_bench.py > gpurun_out/stp_pmc_$kern/out$n.txt 2> gpurun_out/stp_pmc_$kern/err$n.txt
  done
  echo "== $kern"
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for fn in glob.glob("gpurun_out/stp_pmc_$kern/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "stp_kernel" in r["Kernel_Name"] or "stw_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} launches {len(v):3d}  mean {sum(v)/len(v):16.1f}")
PY
  tail -2 gpurun_out/stp_pmc_$kern/err2.txt | cut -c1-300
done
