#!/bin/bash
# Round-5 evidence set on the GPU box, produced ONCE on the round's last tree: tools/profile_round.sh (the driver's bench
# line, rocprofv3 kernel stats of the same command and of c3, every preset, both training steps, the 2-rank line) plus:
# the ResNet training line, the same-box A/B of whole plans with and without the wave-specialised tiles (GV_NO_WS=1),
# warm-repeat vs in-sequence time of every launch of c3 and c5 (with the tile each launch runs), the per-kernel PMC
# summaries of c2, c3 and c5, and the wave-specialised kernel's probes (per layer shape against every other tile; its
# workgroup phase by phase; its ablations) — the last three need the profiling build for the phase table only.
# Usage: bash tools/profile_round_r5.sh TAG     (writes gpurun_out/prof_TAG/ and gpurun_out/pmc_TAG_{c2,c3,c5}/)
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cd $R
python3 bench.py --train --preset c4 > $O/bench_train_c4_bf16.json 2>> $O/bench.err
bash tools/ws_bench_ab.sh "c2 c3 c5 c4" > $O/ws_plan_ab.txt 2>&1
python3 tools/seq_vs_warm.py --preset c3 > $O/seq_vs_warm_c3.txt 2>&1
python3 tools/seq_vs_warm.py --preset c5 > $O/seq_vs_warm_c5.txt 2>&1
python3 tools/ws_probe.py bf16 both > $O/ws_probe.txt 2>&1
bash tools/ws_ablate.sh > $O/ws_ablation.txt 2>&1
# the three-plane twin (csrc/conv_ws_x3.hip): per layer shape against every LDS-DMA tile, its ablations, the GEMM mode
# against the register-staged tiles, and the whole c2 plan launch by launch without and with both kernels' tiles
python3 tools/ws_x3_probe.py 0 > $O/x3ws_probe.txt 2>&1
python3 tools/ws_x3_ablate.py > $O/x3ws_ablation.txt 2>&1
python3 tools/wsg_x3_probe.py 0 16384 > $O/x3wsg_probe.txt 2>&1
GV_NO_WS=1 python3 tools/seq_vs_warm.py --preset c2 > $O/x3ws_seq_vs_warm_c2_without.txt 2>&1
python3 tools/seq_vs_warm.py --preset c2 > $O/x3ws_seq_vs_warm_c2_with.txt 2>&1
bash tools/ws_epi_ab.sh > $O/ws_epilogue_ab.txt 2>&1
[ -f gvcnn-tf_amd/libgvcnn_hip_pt.so ] && python3 tools/ws_phase_times.py bf16 > $O/ws_phase_times.txt 2>&1
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c2 > /dev/null 2>&1
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c3 --preset c3 > /dev/null 2>&1
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c5 --preset c5 > /dev/null 2>&1
cat $O/ws_plan_ab.txt
tail -2 $O/seq_vs_warm_c3.txt
tail -2 $O/seq_vs_warm_c5.txt
for f in bench bench_c3 bench_c4 bench_c5 bench_train_c3_bf16 bench_train_c4_bf16; do echo "== $f"; cut -c1-400 $O/$f.json; echo; done
