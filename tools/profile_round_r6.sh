#!/bin/bash
# Round-6 evidence set on the GPU box, produced ONCE on the round's last tree: tools/profile_round.sh (the driver's bench
# line, rocprofv3 kernel stats of the same command and of c3, every preset, both training steps, the 2-rank gloo line) plus:
# the ResNet training line; the same-box A/B of the whole c4 plan — separate launches (GV_NO_CHAIN=1), conv3 + next preact +
# conv1 as one launch (GV_NO_UNIT=1), one launch per bottleneck unit everywhere (GV_UNIT_ALL=1), the default (units at d = 64,
# the chain at d = 128) — as bench lines and launch by launch; the
# bottleneck launch alone beside the launches it replaces; warm-repeat vs in-sequence tables of c3 / c5; the MFMA-shape
# microbenchmark; the one-rank RCCL test's report; per-kernel PMC summaries of c2 and c4 and, from the same counter passes,
# every launch's HBM traffic against its algorithmic / stored bytes (tools/traffic_per_launch.py); the FETCH_SIZE calibration
# on this library's read patterns (tools/r6_calib.sh); the ResNet stem launch with and without the folded pre-activation
# (tools/r6_poolact.sh) and the first unit's projection shortcut inside / outside its conv3 GEMM (tools/r6_proj.sh); per-launch
# tables of both training steps (tools/step_times.py).
# Usage: bash tools/profile_round_r6.sh TAG     (writes gpurun_out/prof_TAG/ and gpurun_out/pmc_TAG_{c2,c4}/)
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
bash $R/tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
cd $R
python3 bench.py --train --preset c4 > $O/bench_train_c4_bf16.json 2>> $O/bench.err
{
for mode in separate chain units_everywhere default; do
  unset GV_NO_CHAIN GV_NO_UNIT GV_UNIT_ALL
  [ $mode = separate ] && export GV_NO_CHAIN=1
  [ $mode = chain ] && export GV_NO_UNIT=1
  [ $mode = units_everywhere ] && export GV_UNIT_ALL=1
  python3 bench.py --preset c4 --no-cpu-baseline --no-traffic --no-exact > $O/ab_c4_$mode.json 2> $O/ab_c4_$mode.err
  python3 - <<PY
import json
d=json.loads(open("$O/ab_c4_$mode.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("c4 $mode: %.0f views/s, %.3f ms/step, conv %.3f ms in %s, %.0f TF/s, frac %.4f, hbm-bound launches %s | stages %s" % (d["value"], d["ms_per_step"], r.get("conv_ms_per_step",0), r.get("kernel","").split(";")[-2].strip() if ";" in r.get("kernel","") else "", r.get("achieved",0), r.get("frac",0), r.get("hbm_bound_launches"), {k: (round(v["ms"],3), round(v["frac"],3)) for k,v in r.get("stages",{}).items()}))
PY
done
unset GV_NO_CHAIN GV_NO_UNIT GV_UNIT_ALL
} > $O/chain_plan_ab.txt 2>&1
GV_NO_CHAIN=1 python3 tools/seq_vs_warm.py --preset c4 > $O/seq_vs_warm_c4_separate.txt 2>&1
GV_NO_UNIT=1 python3 tools/seq_vs_warm.py --preset c4 > $O/seq_vs_warm_c4_chain.txt 2>&1
GV_UNIT_ALL=1 python3 tools/seq_vs_warm.py --preset c4 > $O/seq_vs_warm_c4_units.txt 2>&1
python3 tools/seq_vs_warm.py --preset c4 > $O/seq_vs_warm_c4_default.txt 2>&1
python3 tools/chain_probe.py > $O/chain_probe.txt 2>&1
python3 tools/seq_vs_warm.py --preset c3 > $O/seq_vs_warm_c3.txt 2>&1
python3 tools/seq_vs_warm.py --preset c5 > $O/seq_vs_warm_c5.txt 2>&1
[ -x gvcnn-tf_amd/build/mfma_shape_ab ] && gvcnn-tf_amd/build/mfma_shape_ab 40000 > $O/mfma_shape_ab.txt 2>&1
python3 -m pytest tests/test_gpu_rccl_one_rank.py -x -q -m gpu -s > $O/rccl_one_rank.txt 2>&1
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c2 > /dev/null 2>&1
bash tools/pmc_bench.sh gpurun_out/pmc_${TAG}_c4 --preset c4 > /dev/null 2>&1
python3 tools/traffic_per_launch.py gpurun_out/pmc_${TAG}_c2 --preset c2 --trace $O/kt/kt_kernel_trace.csv > $O/traffic_per_launch_c2.txt 2>&1
python3 tools/traffic_per_launch.py gpurun_out/pmc_${TAG}_c4 --preset c4 > $O/traffic_per_launch_c4.txt 2>&1
bash tools/r6_poolact.sh > $O/pool_act_ab.log 2>&1; cp gpurun_out/r6/pool_act_ab.txt $O/pool_act_ab.txt
bash tools/r6_proj.sh > $O/proj_ab.log 2>&1; cp gpurun_out/r6/proj_ab.txt $O/proj_ab.txt
[ -x gvcnn-tf_amd/build/fetch_calib ] && bash tools/r6_calib.sh calib-only > $O/fetch_calib.log 2>&1 && cp gpurun_out/r6/fetch_calib.txt $O/fetch_calib.txt
python3 tools/step_times.py --backbone resnet_v2_50 --tune > $O/step_times_train_c4_bf16.txt 2>&1
python3 tools/step_times.py --tune > $O/step_times_train_c3_bf16.txt 2>&1
head -4 $O/traffic_per_launch_c2.txt; head -4 $O/traffic_per_launch_c4.txt; cat $O/pool_act_ab.txt; cat $O/proj_ab.txt
cat $O/chain_plan_ab.txt
cat $O/chain_probe.txt | grep "^d "
tail -2 $O/seq_vs_warm_c3.txt
tail -2 $O/seq_vs_warm_c5.txt
for f in bench bench_c3 bench_c4 bench_c5 bench_train_c3_bf16 bench_train_c4_bf16; do echo "== $f"; cut -c1-400 $O/$f.json; echo; done
