"""Behavior_policy (hard update = the iPLAN-Hard ablation) -- mirror of nova/behavior_policy.py:13-248.

Same networks and kernels as the soft-update policy (nova/stable_behavior_policy.py); the differences
live in two kernel switches: ``latent_update`` takes the encoder output as the new latent (no blending
with the previous one, :78-115) and ``learn`` walks non-overlapping windows with one global loss
normaliser (:119-215).  ``learn`` returns the list of per-agent losses exactly like the reference
(run_ippo.py:271 unpacks three values from it -- a quirk of the reference, reproduced, not fixed).
"""
import numpy as np
import torch

from .. import ops
from ..optim import step_all
from .prediction_policy import _as_dev
from .stable_behavior_policy import Behavior_policy as _SoftBehaviorPolicy


class Behavior_policy(_SoftBehaviorPolicy):
    learn_takes_prepared = False

    def latent_update(self, history, encoder_hidden, prev_latent, out_latent=None, out_hidden=None):
        """nova/behavior_policy.py:78-115: new latent = softmax(encoder(history)), prev_latent is ignored."""
        as_np = isinstance(history, np.ndarray)
        hist = _as_dev(history, self.device)
        hid = _as_dev(encoder_hidden, self.device)
        lat, hL = ops.enc_forward(self.enc_arena, hist.permute(1, 0, 2, 3, 4), hid[:, 0].permute(1, 0, 2, 3), None, 0.0,
                                  self.latent_dim, out_lat=None if out_latent is None else out_latent.permute(1, 0, 2, 3),
                                  out_h=None if out_hidden is None else out_hidden.permute(1, 0, 2, 3))
        lat = lat.permute(1, 0, 2, 3)
        hL = hL.permute(1, 0, 2, 3).unsqueeze(1)
        if as_np:
            return lat.cpu().numpy(), hL
        return lat, hL

    def learn(self, batch, t_env, keep=None):
        """nova/behavior_policy.py:119-215 for all agents at once (episode_limit must be a multiple of
        max_history_len, as the reference's reshape at :142 requires)."""
        a = self.args
        dev = self.device
        history = batch["history"][:, :-1].to(device=dev, dtype=torch.float32)
        term = batch["terminated"][:, :-1].to(dev)
        if history.shape[1] % self.max_history_len:
            raise RuntimeError("hard-update behaviour learning needs episode_limit % max_history_len == 0 "
                               "(nova/behavior_policy.py:142)")
        mask = (1 - term[..., 0]) if a.env == "MPE" else term[..., 0]
        mask = mask.permute(2, 0, 1).to(torch.float32).contiguous()
        hist = history.permute(2, 0, 1, 3, 4)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if keep is None else 0
        fwd = ops.beh_forward(self.enc_arena, self.dec_arena, hist, mask, self.max_history_len, self.latent_dim, 1.0, 0.0,
                              a.decoder_dropout, keep=keep, seed=seed, hard=True,
                              win_norm=self._global_window_sums(mask, hard=True))
        ops.beh_backward(self.enc_arena, self.dec_arena, fwd)
        if getattr(self, "dp", None) is not None:
            self.dp.all_reduce_grads(self.enc_arena, self.dec_arena)
        sq = step_all(self.behavior_optimizer, self.max_grad_norm if self._use_max_grad_norm else None)
        nA = self.n_agents
        host = torch.cat([fwd["loss"][:, 0], sq.sqrt().reshape(-1)]).cpu()
        loss, norms = host[:nA].numpy(), host[nA:].reshape(nA, 2)
        train_info = {"behavior_loss": float(loss.sum()), "behavior_encoder_grad_norm": float(norms[:, 0].sum()),
                      "behavior_decoder_grad_norm": float(norms[:, 1].sum())}
        if t_env - self.log_stats_t >= self.args.learner_log_interval:
            for k, v in train_info.items():
                self.logger.log_stat(self.log_prefix + k, v, t_env)
        return [np.asarray(x) for x in loss]
