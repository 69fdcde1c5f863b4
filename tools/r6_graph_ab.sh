# Round 6: bench.py with and without --graph (one hipGraph per step), alternating twice per preset on one box -> profiles/r6_graph_ab.txt
mkdir -p gpurun_out/r6
for p in c3 c4 c2; do for g in "" "--graph" "" "--graph"; do python bench.py --preset $p --no-cpu-baseline --no-traffic --no-exact $g 2> gpurun_out/r6/g.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$p', '$g' or 'stream', d['value'], d['ms_per_step'])"; done; done
