#!/bin/bash
# Round profile on the GPU box: the driver's bench line, rocprofv3 kernel stats of the same command (single launch lane, so
# kernel durations are comparable with bench.py's per-launch hipEvent timing), the other presets, the training steps.
# Usage: bash tools/profile_round.sh TAG      (writes gpurun_out/prof_TAG/, copy what is to be judged into profiles/)
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --tile-cache $O/tiles_f32.json --cpu-seconds 10 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --no-lanes --tile-cache $O/tiles_f32.json --no-cpu-baseline --no-exact --no-traffic --steps 10 > $O/bench_under_rocprof_nolanes.json 2> $O/kt.err
python3 $R/bench.py --p3 none --no-cpu-baseline --no-exact --no-traffic > $O/bench_p3_none.json 2>> $O/bench.err
for p in c3 c4 c5; do
  python3 $R/bench.py --preset $p --tile-cache $O/tiles_$p.json --no-cpu-baseline --no-exact > $O/bench_$p.json 2>> $O/bench.err
done
rocprofv3 --kernel-trace --stats -d $O/kt16 -o kt --output-format csv -- python3 $R/bench.py --preset c3 --no-lanes --tile-cache $O/tiles_c3.json --no-cpu-baseline --no-exact --no-traffic --steps 10 > $O/bench_c3_under_rocprof_nolanes.json 2> $O/kt16.err
python3 $R/bench.py --train --preset c3 > $O/bench_train_c3_bf16.json 2>> $O/bench.err
python3 $R/bench.py --train > $O/bench_train_f32.json 2>> $O/bench.err
python3 $R/bench.py --gpus 2 --backend gloo --same-device --shapes 8 --steps 5 --warmup 2 --no-roofline > $O/bench_2rank_gloo.json 2>> $O/bench.err
ls -la $O
for f in bench bench_p3_none bench_c3 bench_c4 bench_c5 bench_train_c3_bf16 bench_train_f32 bench_2rank_gloo; do echo "== $f"; cut -c1-700 $O/$f.json; echo; done
