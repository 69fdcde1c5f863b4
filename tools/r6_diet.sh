# Round 6: the wave-specialised kernel's tap-table consumers (GV_WS_DIET=1) against the arithmetic form: parity tests, whole plans
# c3 / c5 alternating twice on one box, and every tile per layer shape with the table form on (profiles/r6_ws_diet_ab.txt).
#   bash tools/r6_diet.sh        (on the GPU box)
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_lowp.py -x -q -m gpu -k "ws" > gpurun_out/r6/t_ws.txt 2>&1; echo "ws rc $?"; tail -n 6 gpurun_out/r6/t_ws.txt
for p in c3 c5; do
for mode in nodiet diet nodiet diet; do
  if [ $mode = nodiet ]; then unset GV_WS_DIET; else export GV_WS_DIET=1; fi
  python bench.py --preset $p --no-cpu-baseline --no-traffic --no-exact > gpurun_out/r6/diet_${p}_${mode}.json 2> gpurun_out/r6/diet_${p}_${mode}.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6/diet_${p}_${mode}.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("${p} ${mode}: %.0f views/s, %.3f ms/step, conv %.3f ms, %.0f TF/s, frac %.4f | stages %s" % (d["value"], d["ms_per_step"], r.get("conv_ms_per_step",0), r.get("achieved",0), r.get("frac",0), {k: round(v["frac"],3) for k,v in r.get("stages",{}).items()}))
PY
done
done 2>&1 | tee gpurun_out/r6/ws_diet_plan_ab.txt

export GV_WS_DIET=1
python tools/ws_probe.py bf16 both > gpurun_out/r6/ws_probe_diet.txt 2>&1; tail -30 gpurun_out/r6/ws_probe_diet.txt
