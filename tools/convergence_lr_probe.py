#!/usr/bin/env python3
"""Loss trajectories of tests/test_gpu_convergence.py's runs at several learning rates (10-step means)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_convergence as T
for bb, S, lrs in (("inception_v3", 96, (0.005, 0.005)), ("resnet_v2_50", 64, (0.004, 0.004))):
    for lr in lrs:
        for storage in ("f32", "bf16"):
            losses, acc = T.run(bb, S, storage, 150, lr, 0.9)
            print(bb, lr, storage, "acc %.3f" % acc, " ".join("%.3f" % np.mean(losses[i:i + 10]) for i in range(0, 150, 10)))
