"""Trainer parity layer, host logic only: the learning-rate policies of utils/train_utils.py:65-118
against hand-evaluated values of tf.train.polynomial_decay / exponential_decay(staircase), and the per-step
protocol (learning rate -> step -> global step -> check_numerics) with a stand-in engine."""
import math

import pytest
import torch

from gvcnn_tf_amd import trainer as T


def test_poly_policy_matches_tf_polynomial_decay():
    # lr = base * (1 - step/total)^power, end_learning_rate = 0, clipped at total
    f = lambda step: T.get_model_learning_rate("poly", 0.001, 0.3, 1e-3, 300000, 0.9, 0, 1e-4, step)
    assert f(0) == 0.001
    assert abs(f(150000) - 0.001 * 0.5 ** 0.9) < 1e-12
    assert abs(f(299999) - 0.001 * (1 / 300000.0) ** 0.9) < 1e-15
    assert f(300000) == 0.0 and f(400000) == 0.0


def test_step_policy_is_staircase_exponential_decay():
    f = lambda step: T.get_model_learning_rate("step", 0.01, 2000, 0.1, 30000, 0.9, 0, 1e-4, step)
    assert f(0) == 0.01 and f(1999) == 0.01
    assert abs(f(2000) - 0.001) < 1e-15 and abs(f(5999) - 0.0001) < 1e-15


def test_slow_start_and_unknown_policy():
    f = lambda step: T.get_model_learning_rate("poly", 0.001, 0.3, 1e-3, 1000, 0.9, 10, 1e-4, step)
    assert f(0) == 1e-4 and f(9) == 1e-4 and f(10) == 0.001 * (1 - 10 / 1000.0) ** 0.9
    with pytest.raises(ValueError):
        T.get_model_learning_rate("cosine", 0.001, 1, 1, 1, 1, 0, 0, 0)


class _Engine:
    backbone = "inception_v3"

    def __init__(self, losses):
        self.losses, self.calls = list(losses), []

    def train_step(self, views, labels, lr, mu, weight_decay):
        self.calls.append((lr, mu, weight_decay))
        return torch.tensor(self.losses[len(self.calls) - 1])


def test_trainer_protocol_and_check_numerics():
    eng = _Engine([2.3, 2.1, float("nan")])
    tr = T.Trainer(eng, training_number_of_steps=100)
    assert tr.weight_decay == 0.00004                              # inception arg scope
    tr.step(None, None)
    tr.step(None, None)
    assert tr.global_step == 2
    assert eng.calls[0][0] == 0.001 and abs(eng.calls[1][0] - 0.001 * 0.99 ** 0.9) < 1e-15 and eng.calls[0][1] == 0.9
    with pytest.raises(FloatingPointError, match="Loss is inf or nan"):
        tr.step(None, None)


class _StateEngine(_Engine):
    """Stand-in with the attributes Trainer.save / restore touch."""
    view_offset = 0

    def __init__(self):
        super().__init__([1.0] * 8)
        g = torch.Generator().manual_seed(0)
        self.params = {"InceptionV3/Conv2d_1a_3x3/weights": torch.randn(3, 3, 3, 32, generator=g),
                       "InceptionV3/Conv2d_1a_3x3/BatchNorm/beta": torch.randn(32, generator=g),
                       "InceptionV3/Conv2d_1a_3x3/BatchNorm/moving_mean": torch.randn(32, generator=g),
                       "dense_2/kernel": torch.randn(2048, 10, generator=g), "dense_2/bias": torch.randn(10, generator=g)}
        self.momentum = {k: torch.randn(v.shape, generator=g) for k, v in self.params.items() if "moving" not in k}
        self.score_kernel = torch.randn(2, 768, generator=g)
        self.score_bias = torch.randn(2, generator=g)
        self._packed_dirty = False


def test_trainer_checkpoint_round_trip(tmp_path):
    eng = _StateEngine()
    tr = T.Trainer(eng, training_number_of_steps=100)
    tr.step(None, None)
    tr.step(None, None)
    want = {k: v.clone() for k, v in eng.params.items()}
    want_m = {k: v.clone() for k, v in eng.momentum.items()}
    sk, sb = eng.score_kernel.clone(), eng.score_bias.clone()
    prefix = str(tmp_path / "model.ckpt-2")
    tr.save(prefix)
    keys = set(tr.state_dict())
    assert {"global_step", "dense/kernel", "dense_1/bias", "dense_2/kernel/Momentum",
            "InceptionV3/Conv2d_1a_3x3/weights/Momentum"} <= keys
    assert "InceptionV3/Conv2d_1a_3x3/BatchNorm/moving_mean/Momentum" not in keys      # not trainable: no slot
    # a fresh trainer with different values restores everything, including the step counter
    eng2 = _StateEngine()
    for d in (eng2.params, eng2.momentum):
        for v in d.values():
            v.add_(1.0)
    eng2.score_kernel.add_(1.0)
    tr2 = T.Trainer(eng2, training_number_of_steps=100)
    assert tr2.restore(prefix) == []
    assert tr2.global_step == 2 and eng2._packed_dirty
    for k in want:
        assert torch.equal(eng2.params[k], want[k]), k
    for k in want_m:
        assert torch.equal(eng2.momentum[k], want_m[k]), k
    assert torch.equal(eng2.score_kernel, sk) and torch.equal(eng2.score_bias, sb)
    assert tr2.learning_rate() == tr.learning_rate()
    # a backbone-only checkpoint (the TF-slim ImageNet case): strict raises, strict=False binds what matches
    from gvcnn_tf_amd import tf_checkpoint
    bb = str(tmp_path / "inception_v3.ckpt")
    tf_checkpoint.write_checkpoint(bb, {k: v.numpy() for k, v in want.items() if k.startswith("InceptionV3/")})
    with pytest.raises(KeyError):
        tr2.restore(bb)
    eng3 = _StateEngine()
    eng3.params["InceptionV3/Conv2d_1a_3x3/weights"].zero_()
    missing = T.Trainer(eng3).restore(bb, strict=False)
    assert torch.equal(eng3.params["InceptionV3/Conv2d_1a_3x3/weights"], want["InceptionV3/Conv2d_1a_3x3/weights"])
    assert "dense_2/kernel" in missing and "global_step" not in missing
