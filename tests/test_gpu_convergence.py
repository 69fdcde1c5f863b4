"""Does a bf16-storage run actually TRAIN like an fp32-storage run?  (the reference's loop:
train.py:259-302 — per step phase 1, host grouping, phase 2 with loss + train_op; validation train.py:319-366.)

The whole-step comparisons of test_gpu_train_lp.py can only be loose: on a randomly initialised network with
train-mode BatchNorm any perturbation grows from layer to layer, so one bf16 step and one fp32 step have nearly
unrelated total gradients (cosine 0.05 at the c3 geometry) although every op agrees with the oracle.  What matters is
whether that is harmless.  Here both engines start from the SAME initialisation and run the same 150 Momentum steps on
a synthetic, learnable multi-view stream (class- and view-dependent mean image + noise, a fresh batch every step);
then the trained variables are bound to the inference engine (moving-average BatchNorm, the `forward` path of
eval.py) and 128 held-out shapes are classified.

Round 3's version of this test failed on the driver's box (fp32 run: loss 2.21 -> 0.09, eval-mode accuracy 0.24 =
chance).  What tools/convergence_diag.py found (gpurun_out of round 4, profiles/r4_convergence_diag.txt):
  * no batch was dropped by the IndexError branch (it is counted and asserted zero now);
  * the SAME trained variables classify the held-out shapes at 0.91 - 0.97 with train-mode BatchNorm (per-view batch
    statistics): the network had learned, the kernels and the inference binding were right (at 128 x 128 input and for
    ResNet the eval-mode engine reached 1.000 with the same code);
  * the eval-mode number swung 0.51 - 1.00 between runs of one engine, for either storage type, because (a) the filter
    gradients were summed with fp32 atomics (two runs differed in the last bits, amplified chaotically over 150 steps),
    (b) at 96 x 96 input Mixed_7's maps are 1 x 1, so a view's batch statistics are taken over 8 values, and (c) the
    moving averages at decay 0.9 are a window over the last ~10 such steps with the last of the V sequential updates
    weighted most — they do not describe the trained network.
Fixed since: (a) training is bitwise reproducible (gv_conv2d_wgrad_ws, tests/test_gpu_wgrad_det.py) and this test
asserts it over the whole run; (b) Inception runs at 128 x 128 (2 x 2 maps in Mixed_7: 32 values per statistic);
(c) the eval-mode check follows a recalibration pass (TrainGVCNN.recalibrate_moving_averages: forward only, fixed
variables, 16 fresh batches) — the stand-in for the reference's decay of 0.9997 over tens of thousands of steps.

Bands.  Measured (profiles/r4_convergence_diag.txt part 2: 8 initialisation / data seeds for Inception, 6 for ResNet, per
storage type; plus this test's own seed): held-out accuracy with train-mode BatchNorm 0.805 - 1.000 (mean 0.95, standard
deviation 0.06), recalibrated eval-mode 0.75 - 1.000 (mean 0.97, s.d. 0.07; ResNet: 1.000 in 11 of 12 runs), plain moving
averages 0.70 - 1.000 (mean 0.945, s.d. 0.10).  The eval-mode numbers of Inception stay the noisier ones for a reason
that is not a kernel's: a view's batch statistics over 8 shapes of varying class mix are part of what the train-mode
network computes, and no fixed statistic reproduces that.  The run is deterministic, so the driver's box computes the
very numbers this box did; the floors are set 5 standard deviations below the measured means so that a change that
merely perturbs the rounding (another draw from the same distribution) still passes:
  * both losses (mean of the last 30 steps) fall below 50 % of their starting level (mean of the first 10; measured
    0.1 - 37 %: the per-batch loss of 8 fresh shapes is noisy);
  * accuracy with train-mode BatchNorm >= 0.60 and recalibrated eval-mode accuracy >= 0.60 for both storage types
    (chance 0.25; one binomial sigma of a 128-shape accuracy at chance is 0.04: the floors sit 9 sigma above chance);
  * no band on bf16 against fp32: two runs' accuracies are two draws with s.d. 0.06 - 0.10, a band tight enough to mean
    something fails by chance (round 3 widened one three times); both must clear the same floors;
  * the plain moving averages are reported, not asserted.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvcnn_tf_amd as gv                          # noqa: E402
from gvcnn_tf_amd import params as gparams          # noqa: E402
from gvcnn_tf_amd.training import TrainGVCNN        # noqa: E402

DEV = "cuda:0"


def make_set(num_shapes, V, S, C, seed, protos):
    """Shapes of class c: views = prototype[c][v] (a smooth random image per class and view) + noise (on the device)."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    labels = torch.randint(0, C, (num_shapes,), generator=g, device=DEV)
    x = protos[labels] + 0.15 * torch.randn(num_shapes, V, S, S, 3, generator=g, device=DEV)
    return x.clamp_(-0.5, 0.5), labels


def prototypes(C, V, S, seed=7):
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(C, V, 8, 8, 3, generator=g)                       # smooth: 8x8 pattern upsampled
    p = torch.nn.functional.interpolate(low.reshape(C * V, 8, 8, 3).permute(0, 3, 1, 2), size=(S, S), mode="bilinear",
                                        align_corners=False).permute(0, 2, 3, 1).reshape(C, V, S, S, 3)
    return (0.3 * p).to(DEV)


def bind_inference(eng, backbone, N, V, S, C, G, storage):
    """The trained variables on the inference path: BatchNorm from the moving statistics, scheme from the device scorer."""
    inf = gv.GVCNN(backbone, N, V, S, S, C, G, device=DEV, num_bins=G, storage=storage)
    inf.plan.bind({k: v.detach().float().cpu() for k, v in eng.params.items() if k not in eng.cls_names})
    H = {}
    for v in range(V):
        kn, bn = gparams.scorer_names(v)
        H[kn], H[bn] = eng.score_kernel[v].cpu().reshape(-1, 1), eng.score_bias[v:v + 1].cpu()
    H[eng.cls_names[0]], H[eng.cls_names[1]] = eng.params[eng.cls_names[0]].cpu(), eng.params[eng.cls_names[1]].cpu()
    inf.set_head(H)
    return inf


def eval_accuracy(inf, test_x, test_y, N):
    correct, dropped = 0, 0
    for b in range(0, test_x.shape[0], N):
        try:
            _, _, logits = inf.forward(test_x[b:b + N].contiguous())
        except IndexError:                              # a score of exactly 1.0: the reference raises here too
            dropped += 1
            continue
        correct += int((logits.argmax(1) == test_y[b:b + N]).sum())
    return correct / test_x.shape[0], dropped


def train(backbone, S, storage, steps, lr, bn_decay, protos, C, V, N, G):
    eng = TrainGVCNN(backbone, N, V, S, S, C, G, device=DEV, num_bins=G, storage=storage, seed=5)
    losses = []
    for it in range(steps):
        xb, yb = make_set(N, V, S, C, 1000 + it, protos)     # a fresh batch every step: nothing to memorise
        eng.forward(xb, yb, check=False)
        eng.backward()
        eng.update_moving_averages(decay=bn_decay)      # (the arg-scope decay 0.997 / 0.9997 needs thousands of steps)
        eng.apply_momentum(lr, 0.9, 1e-4)
        losses.append(eng.loss.clone())
    return eng, torch.cat(losses)


def run(backbone, S, storage, steps, lr, bn_decay, check_repeat=False):
    C, V, N, G = 4, 4, 8, 5
    protos = prototypes(C, V, S)
    test_x, test_y = make_set(128, V, S, C, 12, protos)
    eng, losses_t = train(backbone, S, storage, steps, lr, bn_decay, protos, C, V, N, G)
    assert bool(torch.isfinite(losses_t).all()), "non-finite loss"
    if check_repeat:                                    # the whole run again: the same bits, step by step
        eng2, losses2 = train(backbone, S, storage, steps, lr, bn_decay, protos, C, V, N, G)
        assert torch.equal(losses_t, losses2), "two runs of the same engine differ from step %d on" % int(
            (losses_t != losses2).nonzero()[0])
        assert torch.equal(eng._flat_p, eng2._flat_p)
        del eng2
    losses = losses_t.cpu().numpy()
    acc_moving, dropped = eval_accuracy(bind_inference(eng, backbone, N, V, S, C, G, storage), test_x, test_y, N)
    # the held-out shapes with train-mode BatchNorm (per-view batch statistics, what the training loss measures)
    correct = 0
    for b in range(0, test_x.shape[0], N):
        _, _, logits, _ = eng.forward(test_x[b:b + N].contiguous(), test_y[b:b + N], check=False)
        correct += int((logits.argmax(1) == test_y[b:b + N]).sum())
    acc_train_mode = correct / test_x.shape[0]
    # recalibration on 16 fresh batches (not the held-out ones), then the eval-mode engine again
    eng.recalibrate_moving_averages(make_set(N, V, S, C, 5000 + k, protos)[0] for k in range(16))
    acc_recal, dropped2 = eval_accuracy(bind_inference(eng, backbone, N, V, S, C, G, storage), test_x, test_y, N)
    return dict(first=float(np.mean(losses[:10])), last=float(np.mean(losses[-30:])), moving=acc_moving,
                train_mode=acc_train_mode, recal=acc_recal, dropped=dropped + dropped2)


@pytest.mark.parametrize("backbone,S,steps,lr", [("resnet_v2_50", 64, 150, 0.004), ("inception_v3", 128, 150, 0.005)])
def test_bf16_run_trains_like_the_fp32_run(backbone, S, steps, lr):
    out = {}
    for storage in ("f32", "bf16"):
        r = out[storage] = run(backbone, S, storage, steps, lr, bn_decay=0.9, check_repeat=(storage == "bf16"))
        print("%s %s: loss %.4f -> %.4f; held-out accuracy: train-mode BatchNorm %.3f, moving averages %.3f, recalibrated "
              "%.3f (batches dropped by IndexError: %d)" % (backbone, storage, r["first"], r["last"], r["train_mode"],
                                                            r["moving"], r["recal"], r["dropped"]))
    for storage, r in out.items():
        assert r["dropped"] == 0, "%s: %d evaluation batches raised IndexError" % (storage, r["dropped"])
        assert r["last"] < 0.5 * r["first"], "%s: loss %.4f -> %.4f" % (storage, r["first"], r["last"])
        assert r["train_mode"] >= 0.60, "%s: accuracy with train-mode BatchNorm %.3f (chance 0.25)" % (storage, r["train_mode"])
        assert r["recal"] >= 0.60, "%s: recalibrated eval-mode accuracy %.3f (chance 0.25)" % (storage, r["recal"])
        # the reference's OWN inference path (eval.py: the plain moving averages update_moving_averages wrote, bound to
        # the inference engine) stays gated: at decay 0.9 they are a window over the last ~10 noisy steps, so the number
        # swings (0.51 - 1.00 over the seeds of profiles/r4_convergence_diag.txt) — but a broken update or a broken
        # binding classifies at chance (round 3's failure), far below this floor
        assert r["moving"] >= 0.40, "%s: eval-mode accuracy with the plain moving averages %.3f (chance 0.25)" % (storage, r["moving"])
