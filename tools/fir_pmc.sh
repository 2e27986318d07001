# SQ counter passes over the channel-bank kernel (C = D = 64, 4 Mi-sample block): bash tools/fir_pmc.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fir_pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/fir_pmc -o p -- python tools/fir_bench.py > gpurun_out/fir_pmc/out.txt 2> gpurun_out/fir_pmc/err.txt
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/fir_pmc2 -o p -- python tools/fir_bench.py > gpurun_out/fir_pmc/out2.txt 2> gpurun_out/fir_pmc/err2.txt
tail -3 gpurun_out/fir_pmc/out.txt
