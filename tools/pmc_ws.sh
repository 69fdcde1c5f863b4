# PMC counters of single convolution launches: wave-specialised tiles against the LDS-DMA kernel (run on the GPU box)
#   bash tools/pmc_ws.sh "1 7 192 192 17 17 640" "0 5 -31"      (shape: KH KW CIN COUT H W NB; tiles: WS index, or -(cfg+1))
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${1:-"1 7 192 192 17 17 640"}
TILES=${2:-"0 5 -31"}
mkdir -p $R/gpurun_out/pmc_ws
for t in $TILES; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_ws/t${t}_p1 -o p --output-format csv -- python3 $R/tools/ws_one.py $t $SHAPE 0 6 > $R/gpurun_out/pmc_ws/t${t}_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_ws/t${t}_p2 -o p --output-format csv -- python3 $R/tools/ws_one.py $t $SHAPE 0 6 > $R/gpurun_out/pmc_ws/t${t}_p2.log 2>&1
done
cd $R
for t in $TILES; do echo "== tile $t"; tail -1 gpurun_out/pmc_ws/t${t}_p1.log; python tools/pmc_summary.py $(find gpurun_out/pmc_ws/t${t}_p1 gpurun_out/pmc_ws/t${t}_p2 -name '*counter_collection.csv') | grep -A17 "conv_ws\|conv_dma"; done
