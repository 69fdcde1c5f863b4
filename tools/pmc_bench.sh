#!/bin/bash
# Per-kernel PMC summary of the bench step (single launch lane, tuned tiles): HBM-side traffic (FETCH_SIZE, WRITE_SIZE:
# separate passes, gfx950 corrections as MI355X_MICROARCH.md prescribes) and matrix-pipe / issue counters.
# Usage: bash tools/pmc_bench.sh OUTDIR [bench flags...]      e.g. bash tools/pmc_bench.sh gpurun_out/pmc_r2 --preset c3
O=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --pmc-child tune --tile-cache $R/$O/tiles.json "$@" > /dev/null 2> $R/$O/tune.err
CMD="python3 $R/bench.py --pmc-child 1 --no-lanes --tile-cache $R/$O/tiles.json --steps 2 --warmup 1 $@"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/f -o p --output-format csv -- $CMD > /dev/null 2> $R/$O/f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/w -o p --output-format csv -- $CMD > /dev/null 2> $R/$O/w.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/$O/s -o p --output-format csv -- $CMD > /dev/null 2> $R/$O/s.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU TCC_HIT_sum TCC_MISS_sum -d $R/$O/l -o p --output-format csv -- $CMD > /dev/null 2> $R/$O/l.err
python3 $R/tools/pmc_bench_summary.py $R/$O 2 | tee $R/$O/summary.txt
