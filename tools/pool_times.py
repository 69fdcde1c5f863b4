import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import gvcnn_tf_amd as gv
from gvcnn_tf_amd import backbones
preset = sys.argv[1]
PRESETS = {"c2": ("inception_v3", 12, 224, "f32", "bf16x3"), "c3": ("inception_v3", 12, 224, "bf16", "f32"), "c5": ("inception_v3", 20, 299, "f16", "f32")}
backbone, V, size, storage, math = PRESETS[preset]
dev = torch.device("cuda:0"); nb = 32 * V
plan = backbones.make_plan(backbone, nb, size, size, dev, math=math, dtype=storage, lanes=False)
plan.bind(gv.params.init_backbone_params(plan.param_shapes(), seed=2, perturb_bn=True))
x = (torch.rand(nb, size, size, 3) - 0.5).to(dev)
plan.run(x)
seq = [min(p, q) for p, q in zip(plan.time_each(x, 10), plan.time_each(x, 10))]
for i, op in enumerate(plan.ops):
    if op["kind"] != "conv":
        xx, y = op["x"], op["y"]
        byt = (xx.npix * xx.c * (6 if xx.p3 else plan.esz) + y.npix * y.c * (6 if y.p3 else plan.esz))
        print("%-40s %-5s in %dx%dx%d -> %dx%dx%d  %.4f ms  %.2f TB/s" % (op["name"][-40:], op["kind"], xx.h, xx.w, xx.c, y.h, y.w, y.c, seq[i], byt / seq[i] / 1e9))
