# Round 6: the first ResNet unit's projection shortcut inside its conv3 GEMM (GV_CHAIN_PROJ) against the shortcut as its own
# launch (GV_NO_PROJ=1): parity tests, then the whole c4 plan alternating twice on one box (profiles/r6_proj_ab.txt).
#   bash tools/r6_proj.sh        (on the GPU box)
mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "projection" > gpurun_out/r6/t_proj.txt 2>&1; echo "tests rc $?"; tail -n 3 gpurun_out/r6/t_proj.txt
for mode in separate proj separate proj; do
  if [ $mode = proj ]; then unset GV_NO_PROJ; else export GV_NO_PROJ=1; fi
  python bench.py --preset c4 --no-cpu-baseline --no-traffic --no-exact > gpurun_out/r6/pj_${mode}.json 2> gpurun_out/r6/pj_${mode}.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6/pj_${mode}.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("c4 ${mode}: %.0f views/s, %.3f ms/step, conv %.3f ms, %.0f TF/s, frac %.4f, hbm-bound launches %s | stages %s" % (d["value"], d["ms_per_step"], r.get("conv_ms_per_step",0), r.get("achieved",0), r.get("frac",0), r.get("hbm_bound_launches"), {k: (round(v["ms"],3), round(v["frac"],3)) for k,v in r.get("stages",{}).items()}))
PY
done 2>&1 | tee gpurun_out/r6/proj_ab.txt
unset GV_NO_PROJ
python tools/seq_vs_warm.py --preset c4 2>&1 | head -9 | cut -c1-200
