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
