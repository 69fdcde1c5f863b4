"""Why did an fp32-storage run train to loss 0.09 and classify held-out shapes at chance in eval mode?
  For R repeated runs of the tests' training run this prints:

  * loss first -> last, the eval-mode (moving-average BatchNorm) accuracy and how many batches raised IndexError;
  * the accuracy of the SAME trained variables with train-mode BatchNorm on the held-out batches (per-view batch
    statistics: what the training loss measured);
  * per BatchNorm layer, the moving mean / variance against the batch statistics of the final variables on held-out
    batches (worst layers);
  * the eval-mode accuracy after a recalibration pass (forward only, fixed variables, K held-out batches, the moving
    statistics REPLACED by the mean of the per-view batch statistics, i.e. what V sequential updates converge to).

usage: python tools/convergence_diag.py [backbone] [S] [storage] [runs] [N] [seed0]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gvcnn_tf_amd as gv                          # noqa: E402
from gvcnn_tf_amd import params as gparams          # noqa: E402
from gvcnn_tf_amd.training import TrainGVCNN        # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_convergence import make_set, prototypes   # noqa: E402

DEV = "cuda:0"


def bind_inference(eng, backbone, N, V, S, C, G, storage):
    inf = gv.GVCNN(backbone, N, V, S, S, C, G, device=DEV, num_bins=G, storage=storage)
    inf.plan.bind({k: v.detach().float().cpu() for k, v in eng.params.items() if k not in eng.cls_names})
    H = {}
    for v in range(V):
        kn, bn = gparams.scorer_names(v)
        H[kn], H[bn] = eng.score_kernel[v].cpu().reshape(-1, 1), eng.score_bias[v:v + 1].cpu()
    H[eng.cls_names[0]], H[eng.cls_names[1]] = eng.params[eng.cls_names[0]].cpu(), eng.params[eng.cls_names[1]].cpu()
    inf.set_head(H)
    return inf


def eval_acc(inf, test_x, test_y, N):
    correct, dropped = 0, 0
    for b in range(0, test_x.shape[0], N):
        try:
            _, _, logits = inf.forward(test_x[b:b + N].contiguous())
        except IndexError:
            dropped += 1
            continue
        correct += int((logits.argmax(1) == test_y[b:b + N]).sum())
    return correct / test_x.shape[0], dropped


def main():
    backbone = sys.argv[1] if len(sys.argv) > 1 else "inception_v3"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    storage = sys.argv[3] if len(sys.argv) > 3 else "f32"
    runs = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    N = int(sys.argv[5]) if len(sys.argv) > 5 else 8
    seed0 = int(sys.argv[6]) if len(sys.argv) > 6 else None      # given: run r uses engine seed seed0 + r (training is
                                                                 # deterministic now, so repeated runs of ONE seed are equal)
    steps, lr, decay = 150, 0.005 if backbone == "inception_v3" else 0.004, 0.9
    C, V, G = 4, 4, 5
    protos = prototypes(C, V, S)
    test_x, test_y = make_set(128, V, S, C, 12, protos)
    for r in range(runs):
        eng = TrainGVCNN(backbone, N, V, S, S, C, G, device=DEV, num_bins=G, storage=storage, seed=5 if seed0 is None else seed0 + r)
        losses = []
        for it in range(steps):
            xb, yb = make_set(N, V, S, C, 1000 + it + (0 if seed0 is None else 7919 * (seed0 + r)), protos)
            eng.forward(xb, yb, check=False)
            eng.backward()
            eng.update_moving_averages(decay=decay)
            eng.apply_momentum(lr, 0.9, 1e-4)
            losses.append(float(eng.loss))
        inf = bind_inference(eng, backbone, N, V, S, C, G, storage)
        acc_eval, dropped = eval_acc(inf, test_x, test_y, N)
        # train-mode BatchNorm on the held-out batches + batch statistics per layer
        bns = [op for op in eng.plan.ops if op["kind"] == "bn"]
        m_sum = [torch.zeros_like(op["stat"]["mean"]) for op in bns]
        v_sum = [torch.zeros_like(op["stat"]["var"]) for op in bns]
        correct, nb = 0, 0
        for b in range(0, test_x.shape[0], N):
            _, _, logits, _ = eng.forward(test_x[b:b + N].contiguous(), test_y[b:b + N], check=False)
            correct += int((logits.argmax(1) == test_y[b:b + N]).sum())
            for i, op in enumerate(bns):
                m_sum[i] += op["stat"]["mean"]
                cnt = N * op["x"].h * op["x"].w
                v_sum[i] += op["stat"]["var"] * (cnt / max(cnt - 1, 1))          # the unbiased estimate the update uses
            nb += 1
        acc_train = correct / test_x.shape[0]
        worst = []
        for i, op in enumerate(bns):
            bm, bv = (m_sum[i] / nb), (v_sum[i] / nb)                  # [V, c]
            mm = eng.params[op["name"] + "/moving_mean"]
            mv = eng.params[op["name"] + "/moving_variance"]
            dm = ((mm[None] - bm).abs() / (bv + op["eps"]).sqrt()).max().item()          # in standard deviations
            spread = ((bm - bm.mean(0, keepdim=True)).abs() / (bv + op["eps"]).sqrt()).max().item()   # view to view
            rv = ((mv[None] + op["eps"]) / (bv + op["eps"]))
            worst.append((dm, spread, rv.min().item(), rv.max().item(), op["name"], op["x"].h, op["x"].c))
        worst.sort(reverse=True)
        # recalibration: moving statistics := mean over views and batches of the batch statistics
        for i, op in enumerate(bns):
            eng.params[op["name"] + "/moving_mean"].copy_((m_sum[i] / nb).mean(0))
            eng.params[op["name"] + "/moving_variance"].copy_((v_sum[i] / nb).mean(0))
        inf2 = bind_inference(eng, backbone, N, V, S, C, G, storage)
        acc_recal, dropped2 = eval_acc(inf2, test_x, test_y, N)
        # ... and with the between-view variance of the means included (the pooled statistic)
        for i, op in enumerate(bns):
            bm, bv = (m_sum[i] / nb), (v_sum[i] / nb)
            mean_all = bm.mean(0)
            eng.params[op["name"] + "/moving_variance"].copy_((bv + bm * bm).mean(0) - mean_all * mean_all)
        inf3 = bind_inference(eng, backbone, N, V, S, C, G, storage)
        acc_pooled, _ = eval_acc(inf3, test_x, test_y, N)
        print("run %d %s %s S=%d N=%d: loss %.3f -> %.3f | eval-mode acc %.3f (dropped batches %d) | train-mode-BN acc %.3f | "
              "recalibrated %.3f (dropped %d) | pooled-variance %.3f" %
              (r, backbone, storage, S, N, np.mean(losses[:10]), np.mean(losses[-30:]), acc_eval, dropped, acc_train,
               acc_recal, dropped2, acc_pooled), flush=True)
        for dm, spread, rlo, rhi, name, h, c in worst[:6]:
            print("    |moving_mean - batch mean| %.2f sd, view-to-view mean spread %.2f sd, moving_var/batch var in "
                  "[%.2f, %.2f]  %s (%dx%d, %d ch)" % (dm, spread, rlo, rhi, name, h, h, c), flush=True)


if __name__ == "__main__":
    main()
