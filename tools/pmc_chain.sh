# PMC counters of the bottleneck chain launch alone (run on the GPU box):  bash tools/pmc_chain.sh [64|128]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=${1:-64}
O=$R/gpurun_out/r6/pmc_chain_d$D
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p1 -o p --output-format csv -- python3 $R/tools/chain_probe.py --d $D --iters 5 --only chain > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -d $O/p2 -o p --output-format csv -- python3 $R/tools/chain_probe.py --d $D --iters 5 --only chain > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/p3 -o p --output-format csv -- python3 $R/tools/chain_probe.py --d $D --iters 5 --only chain > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p4 -o p --output-format csv -- python3 $R/tools/chain_probe.py --d $D --iters 5 --only chain > $O/p4.log 2>&1
cd $R
tail -1 $O/p1.log
python tools/pmc_summary.py $(find $O -name '*counter_collection.csv') | grep -A40 "conv_chain"
