cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/psd_pmc
LOG2L=28 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/psd_pmc -o p -- python tools/psd_knob.py > gpurun_out/psd_pmc/out.txt 2> gpurun_out/psd_pmc/err.txt
LOG2L=28 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d gpurun_out/psd_pmc2 -o p -- python tools/psd_knob.py > gpurun_out/psd_pmc/out2.txt 2> gpurun_out/psd_pmc/err2.txt
tail -3 gpurun_out/psd_pmc/out.txt; tail -5 gpurun_out/psd_pmc/err.txt
