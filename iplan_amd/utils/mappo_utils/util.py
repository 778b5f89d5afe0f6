"""Helpers of utils/mappo_utils/util.py that the hot path's callers import by name."""
import math

import numpy as np
import torch


def init(module, weight_init, bias_init, gain=1):
    """utils/mappo_utils/util.py:7-10."""
    weight_init(module.weight.data, gain=gain)
    bias_init(module.bias.data)
    return module


def check(value):
    """numpy -> torch passthrough (util.py:15-17)."""
    return torch.from_numpy(value) if isinstance(value, np.ndarray) else value


def get_grad_norm(params):
    """util.py:19-25."""
    total = 0.0
    for p in params:
        if p.grad is not None:
            total += float(p.grad.norm()) ** 2
    return math.sqrt(total)


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """util.py:27-31."""
    lr = initial_lr - (initial_lr * (epoch / float(total_num_epochs)))
    for group in optimizer.param_groups:
        group["lr"] = lr


def huber_loss(e, d):
    """util.py:33-36 -- deliberately one-sided (large negative errors contribute 0)."""
    return (e.abs() <= d).float() * e ** 2 / 2 + (e > d).float() * d * (e.abs() - d / 2)


def mse_loss(e):
    return e ** 2 / 2
