cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for op in 58 2; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_lp/op${op}_p1 -o p --output-format csv -- python3 $R/tools/one_conv.py --storage bf16 --op $op --reps 5 > $R/gpurun_out/pmc_lp/op${op}_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_lp/op${op}_p2 -o p --output-format csv -- python3 $R/tools/one_conv.py --storage bf16 --op $op --reps 5 > $R/gpurun_out/pmc_lp/op${op}_p2.log 2>&1
done
cd $R
for op in 58 2; do echo "== op $op"; tail -1 gpurun_out/pmc_lp/op${op}_p1.log; python tools/pmc_summary.py $(find gpurun_out/pmc_lp/op${op}_p1 gpurun_out/pmc_lp/op${op}_p2 -name '*counter_collection.csv') | grep -A20 conv_igemm_lp; done
