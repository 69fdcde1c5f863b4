import sys, torch
sys.path.insert(0,'/root/repo')
from gvcnn_tf_amd.training import TrainGVCNN
for name, kw, N, V, S in (("c5-geometry bf16", dict(storage="bf16"), 8, 20, 299), ("f16 storage", dict(storage="f16"), 4, 12, 224),
                          ("resnet bf16 per-shape", dict(storage="bf16", per_shape=True), 8, 12, 224)):
    bb = "resnet_v2_50" if "resnet" in name else "inception_v3"
    eng = TrainGVCNN(bb, N, V, S, S, 40, 10, device="cuda:0", **kw)
    x = (torch.rand(N, V, S, S, 3, generator=torch.Generator().manual_seed(0)) - 0.5).cuda()
    y = torch.randint(0, 40, (N,), generator=torch.Generator().manual_seed(1)).cuda()
    losses = [float(eng.train_step(x, y, lr=1e-3)) for _ in range(4)]
    g = eng.backward()
    fin = all(bool(torch.isfinite(v).all()) for v in g.values())
    print(name, "losses", ["%.4f" % l for l in losses], "finite grads", fin, "mem GB %.1f" % (torch.cuda.max_memory_allocated() / 1e9))
    del eng; torch.cuda.empty_cache()
