# Round 6: the ResNet stem launch (conv1 -> pool1 -> first pre-activation): parity tests, then the whole c4 plan with the
# pre-activation folded (default) and as its own pass (GV_NO_POOL_ACT=1), alternating on one box.
#   bash tools/r6_poolact.sh        (on the GPU box)
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_lowp.py -x -q -m gpu -k "maxpool or max_pool" > gpurun_out/r6/t_poolact.txt 2>&1; echo "tests rc $?"; tail -n 6 gpurun_out/r6/t_poolact.txt
for mode in separate folded separate folded; do
  if [ $mode = folded ]; then unset GV_NO_POOL_ACT; else export GV_NO_POOL_ACT=1; fi
  python bench.py --preset c4 --no-cpu-baseline --no-traffic --no-exact > gpurun_out/r6/pa_${mode}.json 2> gpurun_out/r6/pa_${mode}.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6/pa_${mode}.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("c4 ${mode}: %.0f views/s, %.3f ms/step, conv %.3f ms, %.0f TF/s, frac %.4f | stages %s" % (d["value"], d["ms_per_step"], r.get("conv_ms_per_step",0), r.get("achieved",0), r.get("frac",0), {k: (round(v["ms"],3), round(v["frac"],3)) for k,v in r.get("stages",{}).items()}))
PY
done 2>&1 | tee gpurun_out/r6/pool_act_ab.txt
unset GV_NO_POOL_ACT
python tools/seq_vs_warm.py --preset c4 2>&1 | head -5
