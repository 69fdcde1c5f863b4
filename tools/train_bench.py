#!/usr/bin/env python3
"""Training-step throughput (--storage f32: fp32 storage, bf16x3 math; bf16: 16-bit storage and MFMA): forward(train BN) + loss + backward + Momentum.
    python tools/train_bench.py [--backbone inception_v3] [--shapes 8] [--views 12] [--size 224]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gvcnn_tf_amd.training import TrainGVCNN

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="inception_v3")
ap.add_argument("--shapes", type=int, default=8)
ap.add_argument("--views", type=int, default=12)
ap.add_argument("--size", type=int, default=224)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--storage", default="f32", choices=["f32", "bf16", "f16"])
ap.add_argument("--graph", action="store_true", help="replay the whole step from one captured graph")
ap.add_argument("--lanes", action="store_true", help="branches of a block on separate streams (meant for --graph)")
a = ap.parse_args()
eng = TrainGVCNN(a.backbone, a.shapes, a.views, a.size, a.size, 40, 10, device="cuda:0", storage=a.storage)
x = (torch.rand(a.shapes, a.views, a.size, a.size, 3) - 0.5).cuda()
labels = torch.randint(0, 40, (a.shapes,)).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
ap_tune = os.environ.get("GV_NO_TUNE") is None
eng.train_step(x, labels, lr=1e-6)
if ap_tune:
    eng.autotune()              # untimed: per-launch tile choice (speed only)
    eng.train_step(x, labels, lr=1e-6)
torch.cuda.synchronize()
tf = tb = to = 0.0
for _ in range(a.steps):
    ev[0].record(); eng.forward(x, labels, check=False); ev[1].record(); eng.backward(); ev[2].record()
    eng.apply_momentum(1e-6); eng.repack(); ev[3].record(); torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2]); to += ev[2].elapsed_time(ev[3])
if a.lanes:
    eng.enable_lanes()
    eng.train_step(x, labels, lr=1e-6)
    torch.cuda.synchronize()
if a.graph:
    # the whole step (forward, loss, backward, moving averages, Momentum, filter re-pack) as ONE graph launch
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        eng.train_step(x, labels, lr=1e-6); eng.repack()
    torch.cuda.current_stream().wait_stream(s_)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        eng.forward(x, labels, check=False); eng.backward(); eng.update_moving_averages(); eng.apply_momentum(1e-6); eng.repack(sync=False)
    gr.replay(); torch.cuda.synchronize()
    ev[0].record()
    for _ in range(a.steps):
        gr.replay()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / a.steps
    print("[%s] graph replay: %.2f ms/step => %.1f views/s (loss %.4f)" % (a.storage, ms, a.shapes * a.views / (ms * 1e-3), float(eng.loss)))
n = a.steps
flops = sum(op.get("flops", 0) for op in eng.plan.ops)
print("[%s] " % a.storage + "%s %dx%d views %d^2: forward %.2f ms, backward %.2f ms, update+repack %.2f ms => %.1f views/s; fwd %.1f TF/s, bwd(2x flops) %.1f TF/s"
      % (a.backbone, a.shapes, a.views, a.size, tf / n, tb / n, to / n, a.shapes * a.views / ((tf + tb + to) / n * 1e-3),
         flops / (tf / n) / 1e9, 2 * flops / (tb / n) / 1e9))
