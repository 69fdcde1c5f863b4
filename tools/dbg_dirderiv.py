import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import gvcnn_tf_amd as gv
from gvcnn_tf_amd.training import TrainGVCNN
backbone, size = sys.argv[1], int(sys.argv[2])
N, V, C_, G = 4, 2, 5, 10
eng = TrainGVCNN(backbone, N, V, size, size, C_, G, device="cuda:0")
P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=5, perturb_bn=True)
Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, C_, seed=6, spread_scores=True)
eng = TrainGVCNN(backbone, N, V, size, size, C_, G, backbone_params=P, head_params=Hd, device="cuda:0")
x = (torch.rand(N, V, size, size, 3, generator=torch.Generator().manual_seed(1)) - 0.5).cuda()
labels = torch.tensor([0, 3, 1, 2])
eng.forward(x, labels)
scheme, weight = eng.scheme.cpu().numpy().copy(), eng.weight.cpu().numpy().copy()
grads = {k: v.clone() for k, v in eng.backward().items()}
w0 = {k: eng.params[k].clone() for k in grads}
def dd(keys, eps):
    gn = float(torch.sqrt(sum((grads[k].double() ** 2).sum() for k in keys)))
    L = []
    for sgn in (1.0, -1.0):
        for k in grads: eng.params[k].copy_(w0[k])
        for k in keys: eng.params[k].copy_(w0[k] + sgn * eps * grads[k] / gn)
        eng._packed_dirty = True
        L.append(float(eng.forward(x, labels, g_scheme=scheme, g_weight=weight)[3]))
    return (L[0] - L[1]) / (2 * eps), gn
allk = list(grads)
for eps in (2e-3, 5e-4, 1e-4):
    print("all eps", eps, dd(allk, eps))
groups = {"cls": [k for k in allk if k.startswith("dense")], "weights": [k for k in allk if k.endswith("/weights")],
          "beta": [k for k in allk if k.endswith("beta")], "gamma": [k for k in allk if k.endswith("gamma")],
          "biases": [k for k in allk if k.endswith("biases")]}
for n, ks in groups.items():
    if ks: print(n, dd(ks, 5e-4))
