"""Trainer parity layer: the host-side schedule around the training step.

    get_model_learning_rate(...)   utils/train_utils.py:65-118 ('step' = staircase exponential decay, 'poly' =
                                   polynomial decay to 0, slow start for the first steps)
    Trainer                        train.py:166-187,259-302: learning rate -> forward/backward/update (BN moving
                                   averages + Momentum) -> global step; `check_numerics` on the loss (train.py:175)

Pure host logic on top of gvcnn-tf_amd/training.py (no arithmetic on tensors here).  Not built: summaries,
checkpoints (tf.train.Saver) and the TF-checkpoint importer.
"""
import math


def get_model_learning_rate(learning_policy, base_learning_rate, learning_rate_decay_step,
                            learning_rate_decay_factor, training_number_of_steps, learning_power,
                            slow_start_step, slow_start_learning_rate, global_step):
    """utils/train_utils.py:65-118 evaluated for one `global_step` (the reference reads the TF global-step variable)."""
    if learning_policy == "step":
        # tf.train.exponential_decay(staircase=True)
        lr = base_learning_rate * learning_rate_decay_factor ** (global_step // learning_rate_decay_step)
    elif learning_policy == "poly":
        # tf.train.polynomial_decay(end_learning_rate=0, cycle=False)
        s = min(global_step, training_number_of_steps)
        lr = base_learning_rate * (1.0 - s / float(training_number_of_steps)) ** learning_power
    else:
        raise ValueError("Unknown learning policy.")
    return slow_start_learning_rate if global_step < slow_start_step else lr


class Trainer:
    """train.py's per-step protocol around a TrainGVCNN (or sharding.ShardedTrainGVCNN) engine.  Defaults are the
    reference's flag defaults (train.py:36-60)."""

    def __init__(self, engine, learning_policy="poly", base_learning_rate=0.001, learning_rate_decay_step=0.3,
                 learning_rate_decay_factor=1e-3, training_number_of_steps=300000, learning_power=0.9,
                 slow_start_step=0, slow_start_learning_rate=1e-4, momentum=0.9, weight_decay=None, check_every=1):
        self.engine = engine
        if weight_decay is None:        # the backbone's arg scope: inception_utils.py:30 (4e-5), resnet_utils.py:198 (1e-4)
            eng = getattr(engine, "eng", engine)
            weight_decay = 0.00004 if getattr(eng, "backbone", "") == "inception_v3" else 0.0001
        self.cfg = (learning_policy, base_learning_rate, learning_rate_decay_step, learning_rate_decay_factor,
                    training_number_of_steps, learning_power, slow_start_step, slow_start_learning_rate)
        self.momentum, self.weight_decay = momentum, weight_decay
        self.check_every = check_every
        self.global_step = 0

    def learning_rate(self):
        return get_model_learning_rate(*self.cfg, self.global_step)

    def step(self, views, labels):
        """One training step; returns the loss (device scalar).  Raises FloatingPointError like
        tf.debugging.check_numerics(total_loss, 'Loss is inf or nan.') (train.py:175) — the check reads the loss back,
        so `check_every` > 1 amortises that synchronisation."""
        loss = self.engine.train_step(views, labels, lr=self.learning_rate(), mu=self.momentum,
                                      weight_decay=self.weight_decay)
        self.global_step += 1
        if self.check_every and self.global_step % self.check_every == 0:
            v = float(loss.item())
            if math.isnan(v) or math.isinf(v):
                raise FloatingPointError("Loss is inf or nan.")
        return loss
