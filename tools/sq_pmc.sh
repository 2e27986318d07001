#!/bin/bash
# SQ counter passes over one kernel:  bash tools/sq_pmc.sh <kernel-name-substring> <out-tag> -- <command ...>
# (two rocprofv3 --pmc passes of 8 SQ counters each + FETCH_SIZE + WRITE_SIZE; per-launch means of the matching kernel)
KERN=$1; TAG=$2; shift 3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/sq_$TAG
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -o p -- "$@" > $O/out1.txt 2> $O/err1.txt
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/p2 -o p -- "$@" > $O/out2.txt 2> $O/err2.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p -- "$@" > $O/out3.txt 2> $O/err3.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p -- "$@" > $O/out4.txt 2> $O/err4.txt
python - "$KERN" "$O" <<'PY'
import csv, glob, collections, sys
kern, o = sys.argv[1], sys.argv[2]
for d in ("p1", "p2", "p3", "p4"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{o}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if kern in r["Kernel_Name"]:
                acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for g, a in sorted(acc.items()):
        for k, v in sorted(a.items()):
            print(f"grid {g:>9s} {k:24s} launches {len(v):3d}  mean {sum(v)/len(v):16.1f}")
PY
tail -3 $O/out1.txt
