"""Scalar schedules of the depth-guided loss (host code, no tensors).

* Decay / ExponentialDecay / LinearDecay / get_depth_scheduler: API of the reference's
  src/depth_decay_modules.py:4-65 (which nothing in the reference calls).
* legacy_decay_step: the decay that is live in the reference's training step
  (src/train_segmentation.py:356-375), including the step-0 sample decay (quirk Q9).
"""
from typing import Union

Number = Union[int, float]


class Decay:
    def __init__(self, init_value: Number, decay_rate: float, update_every: int, min_value: Number):
        if not decay_rate > 0:
            raise AssertionError("Decay rate must be positive")
        if type(init_value) != type(min_value):
            raise AssertionError("Init value and min value must be of the same type")
        self.init_value, self.decay_rate = init_value, decay_rate
        self.update_every, self.min_value = update_every, min_value
        self.return_type = type(init_value)

    def calculate(self, step: int):
        raise NotImplementedError

    def return_update(self, step: int):
        k = step // self.update_every
        if k == 0:
            return self.init_value
        value = self.calculate(k)
        return value if type(value) == self.return_type else self.return_type(value)


class ExponentialDecay(Decay):
    def calculate(self, step: int):
        if type(step) != int:
            raise AssertionError("Step must be an integer")
        return max(self.init_value * self.decay_rate ** step, self.min_value)


class LinearDecay(Decay):
    def calculate(self, step: int):
        if type(step) != int:
            raise AssertionError("Step must be an integer")
        return max(self.init_value - step * self.decay_rate, self.min_value)


class StepDecay(Decay):
    pass


def get_depth_scheduler(version: str):
    table = {"exp": ExponentialDecay, "lin": LinearDecay}
    if version not in table:
        raise NotImplementedError
    return table[version]


def legacy_decay_step(cfg, loss_cfg, global_step: int) -> None:
    """Apply the per-step cfg mutations of src/train_segmentation.py:356-375 for `global_step`."""
    if cfg.depth_loss_decay and global_step > 0 and global_step % cfg.decay_every_steps == 0:
        cfg.depth_feat_weight = cfg.depth_feat_weight * cfg.depth_loss_decay_factor
        if not cfg.fix_depth_feat_shift:
            cfg.depth_feat_shift = cfg.depth_feat_shift * cfg.depth_loss_decay_factor
    if cfg.fps_until_step > 0 and global_step >= cfg.fps_until_step:
        loss_cfg.depth_sampling = "none"
        loss_cfg.feature_samples = cfg.post_fps_samples
    if cfg.fps_sample_decay and global_step % cfg.fps_sample_decay_every_steps == 0:   # fires at step 0 (Q9)
        loss_cfg.feature_samples = max(int(loss_cfg.feature_samples * cfg.fps_sample_decay_factor),
                                       cfg.fps_min_samples)
