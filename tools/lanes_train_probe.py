import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from gvcnn_tf_amd.training import TrainGVCNN
bb = sys.argv[1] if len(sys.argv) > 1 else "inception_v3"
eng = TrainGVCNN(bb, 32, 12, 224, 224, 40, 7, device="cuda:0", num_bins=7, storage="bf16")
x = (torch.rand(32, 12, 224, 224, 3) - 0.5).cuda(); labels = torch.randint(0, 40, (32,)).cuda()
eng.train_step(x, labels, lr=1e-6); eng.autotune(); eng.train_step(x, labels, lr=1e-6)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
step = lambda: eng.train_step(x, labels, lr=1e-6)
print(bb, "eager single stream: %.2f ms" % timed(step))
eng.enable_lanes()
step()
print(bb, "eager, branch lanes: %.2f ms" % timed(step))
eng._lane_streams = None
print(bb, "eager single stream again: %.2f ms" % timed(step))
