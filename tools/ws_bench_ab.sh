# Same-box A/B of whole plans: the autotuner without (GV_NO_WS=1) and with the wave-specialised tiles.
#   bash tools/ws_bench_ab.sh "c3 c5 c4"
mkdir -p gpurun_out/r5
for p in ${1:-c3 c5}; do
  for mode in nows ws; do
    if [ $mode = nows ]; then export GV_NO_WS=1; else unset GV_NO_WS; fi
    python bench.py --preset $p --no-cpu-baseline --no-traffic --no-exact > gpurun_out/r5/ab_${p}_${mode}.json 2> gpurun_out/r5/ab_${p}_${mode}.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r5/ab_${p}_${mode}.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("${p} ${mode}: %.0f views/s, %.3f ms/step, conv %.3f ms, %.0f TF/s, frac %.4f | stages %s" % (d["value"], d["ms_per_step"], r.get("conv_ms_per_step",0), r.get("achieved",0), r.get("frac",0), {k: round(v["frac"],3) for k,v in r.get("stages",{}).items()}))
PY
  done
done
