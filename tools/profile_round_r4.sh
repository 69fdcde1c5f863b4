#!/bin/bash
# Round-4 evidence set on the GPU box, produced ONCE on the round's last tree: tools/profile_round.sh (the driver's bench
# line, rocprofv3 kernel stats of the same command and of c3, every preset, both training steps, the 2-rank line) plus:
# the ResNet-v2-50 training line (configs[3]'s backbone), the in-sequence per-launch table of the bf16 training step and
# its rocprofv3 family summary (filter-gradient kernels and their slice-reduce launches listed apart), warm-repeat vs
# in-sequence time of every launch of c3, and the per-kernel PMC summaries of c2 and c3.
# Usage: bash tools/profile_round_r4.sh TAG     (writes gpurun_out/prof_TAG/ and gpurun_out/pmc_TAG_{c2,c3}/)
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cd $R
python3 bench.py --train --preset c4 > $O/bench_train_c4_bf16.json 2>> $O/bench.err
python3 tools/step_times.py --tune > $O/step_times_train_c3_bf16.txt 2>&1
python3 tools/step_times.py --tune --backbone resnet_v2_50 > $O/step_times_train_c4_bf16.txt 2>&1
python3 tools/seq_vs_warm.py --preset c3 > $O/seq_vs_warm_c3.txt 2>&1
GRAFT_REPO_ROOT=$R bash tools/profile_train.sh bf16 1 > $O/profile_train_bf16.log 2>&1
cp $R/gpurun_out/prof_train_bf16/summary.txt $O/train_bf16_summary.txt 2>/dev/null
cp $R/gpurun_out/prof_train_bf16/kt/kt_kernel_stats.csv $O/train_bf16_kernel_stats.csv 2>/dev/null
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c2 > /dev/null 2>&1
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c3 --preset c3 > /dev/null 2>&1
tail -1 $O/step_times_train_c3_bf16.txt
tail -1 $O/step_times_train_c4_bf16.txt
tail -2 $O/seq_vs_warm_c3.txt
for f in bench bench_c3 bench_c4 bench_c5 bench_train_c3_bf16 bench_train_c4_bf16; do echo "== $f"; cut -c1-400 $O/$f.json; echo; done
