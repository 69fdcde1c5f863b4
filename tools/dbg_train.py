import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import gvcnn_tf_amd as gv
from gvcnn_tf_amd.training import TrainGVCNN
from oracle import train as OT
backbone, size, N, V = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), 2
eng = TrainGVCNN(backbone, N, V, size, size, 5, 10, device="cuda:0")
P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, 5, seed=3, spread_scores=True)
eng = TrainGVCNN(backbone, N, V, size, size, 5, 10, backbone_params=P, head_params=Hd, device="cuda:0")
x = torch.rand(N, V, size, size, 3, generator=torch.Generator().manual_seed(0)) - 0.5
labels = torch.tensor([1, 4, 2, 0, 3, 1][:N])
ref = OT.loss_and_grads(x, labels.numpy(), P, Hd, 10, backbone)
eng.forward(x.cuda(), labels); grads = eng.backward(); torch.cuda.synchronize()
F = eng.view(eng.final).cpu().numpy().reshape(N, V, *eng.view(eng.final).shape[1:])
dF = eng.view(eng.final, grad=True).cpu().numpy().reshape(F.shape)
for lo, hi in ((0,320),(320,704),(704,1088),(1088,1472),(1472,1856),(1856,2048)):
    fo = np.stack([f.numpy() for f in ref['finals']], 1)[..., lo:hi]; go = np.stack([g.numpy() for g in ref['final_grads']], 1)[..., lo:hi]
    print('slice', lo, hi, 'fwd rel', np.abs(F[..., lo:hi]-fo).max()/np.abs(fo).max(), 'dF rel', np.abs(dF[..., lo:hi]-go).max()/np.abs(go).max(), 'mask mismatch', int(((F[..., lo:hi]>0)!=(fo>0)).sum()))
l2=[]; nrm=[]
for nm, gref in ref["grads"].items():
    a, d = grads[nm].cpu().numpy().astype(np.float64), gref.numpy().astype(np.float64)
    l2.append((np.linalg.norm(a-d), np.linalg.norm(d), nm))
big = max(n for _, n, _ in l2)
sig = [(e/n, nm) for e, n, nm in l2 if n > 1e-3*big]
sig.sort(reverse=True)
print("tensors %d significant %d; L2 rel max %.3e median %.3e worst %s" % (len(l2), len(sig), sig[0][0], sig[len(sig)//2][0], [(round(v,4), n[-44:]) for v,n in sig[:3]]))
small = [(e/big, nm) for e, n, nm in l2 if n <= 1e-3*big]
print("insignificant: max abs err / big = %.3e" % (max(v for v,_ in small) if small else 0))
