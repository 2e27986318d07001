#!/bin/bash
# a second set of SQ counters (VMEM / LDS issue):  bash tools/sq_pmc2.sh <kernel-substring> <tag> -- <command ...>
KERN=$1; TAG=$2; shift 3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/sq2_$TAG
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $O/p1 -o p -- "$@" > $O/out1.txt 2> $O/err1.txt
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/p2 -o p -- "$@" > $O/out2.txt 2> $O/err2.txt
python - "$KERN" "$O" <<'PY'
import csv, glob, collections, sys
kern, o = sys.argv[1], sys.argv[2]
for d in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{o}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if kern in r["Kernel_Name"]:
                acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g, a in sorted(acc.items()):
        for k, v in sorted(a.items()):
            print(f"grid {g:>9s} {k:24s} launches {len(v):3d}  mean {sum(v)/len(v):16.1f}")
PY
tail -2 $O/out1.txt
