#!/bin/bash
# Round profile on the GPU box: tuned bench, rocprofv3 kernel stats (single launch lane, so kernel durations are
# comparable with bench.py's per-launch hipEvent timing), HBM traffic PMC passes.  Usage: bash tools/profile_round.sh TAG
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --tile-cache $O/tiles_f32.json --cpu-seconds 12 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --no-lanes --tile-cache $O/tiles_f32.json --no-cpu-baseline --no-exact --steps 10 > $O/bench_under_rocprof_nolanes.json 2> $O/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 $R/bench.py --no-lanes --tile-cache $O/tiles_f32.json --no-cpu-baseline --no-exact --no-roofline --steps 3 --warmup 1 > /dev/null 2> $O/pmc_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 $R/bench.py --no-lanes --tile-cache $O/tiles_f32.json --no-cpu-baseline --no-exact --no-roofline --steps 3 --warmup 1 > /dev/null 2> $O/pmc_w.err
python3 $R/tools/traffic_from_pmc.py $(find $O/pmc_f -name '*counter_collection.csv') $(find $O/pmc_w -name '*counter_collection.csv') 3 > $O/traffic_pmc.json
# 16-bit storage (configs c3-c5 dtype): bench + kernel stats
python3 $R/bench.py --preset c3 --tile-cache $O/tiles_bf16.json --no-cpu-baseline --no-exact > $O/bench_c3_bf16.json 2>> $O/bench.err
python3 $R/bench.py --preset c4 --no-cpu-baseline --no-exact > $O/bench_c4_bf16.json 2>> $O/bench.err
python3 $R/bench.py --preset c5 --no-cpu-baseline --no-exact > $O/bench_c5_f16.json 2>> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/kt16 -o kt --output-format csv -- python3 $R/bench.py --preset c3 --no-lanes --tile-cache $O/tiles_bf16.json --no-cpu-baseline --no-exact --steps 10 > $O/bench_c3_under_rocprof_nolanes.json 2> $O/kt16.err
ls -la $O
cat $O/traffic_pmc.json
