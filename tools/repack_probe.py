import sys; sys.path.insert(0, "/root/repo")
import torch
from gvcnn_tf_amd.training import TrainGVCNN
for bb in ("inception_v3", "resnet_v2_50"):
    eng = TrainGVCNN(bb, 2, 12, 224, 224, 40, 7, device="cuda:0", num_bins=7, storage="bf16")
    eng.repack(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.repack(sync=False)
    e1.record(); e1.synchronize()
    print(bb, "repack %.4f ms" % (e0.elapsed_time(e1) / 20))
