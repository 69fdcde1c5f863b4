#!/usr/bin/env python3
"""Which part of the multi-stream training step survives graph capture on this HIP runtime (each case in a child
process: a failing capture takes the process down)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ["fwd_ops:60", "fwd_ops:64", "fwd_backbone", "fwd_bwd_all"]
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=300)
        print(c, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1][:200])
    sys.exit(0)
import torch
from gvcnn_tf_amd.training import TrainGVCNN
case = sys.argv[1]
storage = os.environ.get("STORAGE", "bf16")
eng = TrainGVCNN("inception_v3", 4, 3, 171, 171, 5, 10, device="cuda:0", storage=storage)
x = (torch.rand(4, 3, 171, 171, 3) - 0.5).cuda()
labels = torch.tensor([1, 4, 2, 0]).cuda()
eng.forward(x, labels, check=False); eng.backward()
eng.enable_lanes(int(os.environ.get("LANES", "3")))
eng.forward(x, labels, check=False); eng.backward(); torch.cuda.synchronize()
def body():
    if case == "fwd_backbone":
        eng.forward_backbone(x)
    elif case == "fwd_all":
        eng.forward(x, labels, check=False)
    elif case == "fwd_bwd_head":
        eng.forward(x, labels, check=False); eng.backward_head()
    elif case == "fwd_bwd_all":
        eng.forward(x, labels, check=False); eng.backward()
    elif case == "bwd_backbone_only":
        eng.backward_head(); eng.backward_backbone()
    elif case.startswith("fwd_ops:"):
        k = int(case.split(":")[1])
        ops = eng.plan.ops
        eng.plan.ops = ops[:k]
        if case.endswith("lane0"):
            for o in ops[57:k]:
                o["lane"] = 0
        if case.endswith("skip57"):
            eng.plan.ops = ops[:57] + ops[59:k]
        try:
            eng.forward_backbone(x)
        finally:
            eng.plan.ops = ops
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay(); torch.cuda.synchronize()
print("captured and replayed, n ops", len(eng.plan.ops))
