#!/bin/bash
# Round-3 evidence set on the GPU box: tools/profile_round.sh (bench lines of every preset, rocprofv3 kernel stats of the
# driver's command and of c3, both training steps, 2-rank line) plus what round 3 added: the in-sequence per-launch table
# of the bf16 training step, rocprofv3 summary of that step, the per-workgroup phase split of the LDS-DMA convolution
# (needs gvcnn-tf_amd/libgvcnn_hip_pt.so: GV_PHASE_TIMES=1 python gvcnn-tf_amd/build.py), the input pipeline's rate, and the
# per-kernel PMC summaries of c2 and c3.
# Usage: bash tools/profile_round_r3.sh TAG     (writes gpurun_out/prof_TAG/ and gpurun_out/pmc_TAG_{c2,c3}/)
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cd $R
python3 tools/step_times.py --tune > $O/step_times_train_c3_bf16.txt 2>&1
[ -f gvcnn-tf_amd/libgvcnn_hip_pt.so ] && python3 tools/phase_times.py > $O/phase_times_bf16.txt 2>&1
python3 tools/pipeline_bench.py --sizes 224 --workers 32 64 --files 1 4 > $O/pipeline_bench.txt 2>&1
python3 tools/conv_probe_lp.py bf16 > $O/conv_probe_bf16.txt 2>&1
GRAFT_REPO_ROOT=$R bash tools/profile_train.sh bf16 1 > $O/profile_train_bf16.log 2>&1
cp $R/gpurun_out/prof_train_bf16/summary.txt $O/train_bf16_summary.txt 2>/dev/null
cp $R/gpurun_out/prof_train_bf16/kt/kt_kernel_stats.csv $O/train_bf16_kernel_stats.csv 2>/dev/null
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c2 > /dev/null 2>&1
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c3 --preset c3 > /dev/null 2>&1
tail -3 $O/step_times_train_c3_bf16.txt
tail -5 $O/pipeline_bench.txt
for f in bench bench_c3 bench_c5 bench_train_c3_bf16; do echo "== $f"; cut -c1-400 $O/$f.json; echo; done
