mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_lowp.py tests/test_gpu_bn_fusion.py -x -q -m gpu > gpurun_out/r6/t_lowp.txt 2>&1; echo "lowp rc $?"; tail -n 4 gpurun_out/r6/t_lowp.txt
for p in c4 c3 c5; do python bench.py --preset $p --no-cpu-baseline --no-traffic --no-exact > gpurun_out/r6/q_$p.json 2> gpurun_out/r6/q_$p.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r6/q_$p.json").read().strip().splitlines()[-1])
r=d.get("roofline",{})
print("$p: %.0f views/s, %.3f ms/step, frac %.4f | stages %s" % (d["value"], d["ms_per_step"], r.get("frac",0), {k: (round(v["ms"],3), round(v["frac"],3)) for k,v in r.get("stages",{}).items()}))
PY
done
python tools/seq_vs_warm.py --preset c4 2>&1 | head -3 | tail -1
python tools/seq_vs_warm.py --preset c3 2>&1 | head -3 | tail -1
