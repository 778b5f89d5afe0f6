"""Behavior_policy of the FC ablation (iPLAN-FC) -- mirror of nova/behavior_FC_policy.py:13-269: three-layer perceptrons
instead of the encoder / decoder GRUs, the window's flattened history as input, no recurrent state and no soft update.

All windows of an episode are independent given the parameters (the decoder of window j only needs the encoder's output
for window j - 1), so ``learn`` is four launches for all agents and all E * N * J rows at once -- encoder forward,
decoder forward (+ L1 numerator), decoder backward (-> d latent), encoder backward -- plus the ``iplan_wgrad``
contractions; window gathers, the one-window latent shift and concatenations are torch data movement.
Reference quirks reproduced: the latent fed to the decoder of window j is the encoder output of window j - 1 (zeros for
j = 0); ``mask_over_next_traj`` is never written by ``behavior_traj_wrapper`` (both loops at :131-136 fill the CURRENT
mask), so the loss is the unmasked mean; ``learn`` returns an empty ``stability_loss`` list."""
import numpy as np
import torch

from .. import ops
from ..arena import ParamArena
from ..optim import FusedAdam, step_all
from .behavior_FC_net import Encoder_3FC, LILI_Latent_Decoder
from .prediction_policy import _as_dev
from .stable_behavior_policy import Behavior_policy as _SoftBehaviorPolicy

EPS = 1e-10


class Behavior_policy(_SoftBehaviorPolicy):
    learn_takes_prepared = False

    def init_behavior_net(self):
        """nova/behavior_FC_policy.py:55-76."""
        a = self.args
        Ld = a.obs_shape_single * a.max_history_len
        self.behavior_encoder = [Encoder_3FC(Ld, a.encoder_rnn_dim, a.latent_dim) for _ in range(self.n_agents)]
        self.behavior_decoder = [LILI_Latent_Decoder(Ld + a.latent_dim, a.decoder_rnn_dim, Ld) for _ in range(self.n_agents)]
        self.enc_arena = ParamArena(self.behavior_encoder, self.device)
        self.dec_arena = ParamArena(self.behavior_decoder, self.device)
        self.behavior_optimizer = [
            FusedAdam([(self.enc_arena, i), (self.dec_arena, i)], lr=a.lr_behavior, eps=self.optim_eps,
                      weight_decay=self.weight_decay) for i in range(self.n_agents)]

    # ---------------------------------------------------------------------------- rollout
    def latent_update(self, history, encoder_hidden=None, prev_latent=None, out_latent=None, out_hidden=None):
        """nova/behavior_FC_policy.py:79-108: latent = Encoder_3FC(flattened window); the hidden state passes through."""
        as_np = isinstance(history, np.ndarray)
        hist = _as_dev(history, self.device).to(torch.float32)
        E, nA, N, L, d = hist.shape
        a = self.args
        x = hist.permute(1, 0, 2, 3, 4).reshape(nA, E * N, L * d).contiguous()
        lat = ops.mlp3_forward(self.enc_arena, "", x, a.encoder_rnn_dim, self.latent_dim, softmax=True, save=False)["out"]
        lat = lat.reshape(nA, E, N, self.latent_dim).permute(1, 0, 2, 3)
        if out_latent is not None:
            out_latent.copy_(lat)
        if as_np:
            return lat.contiguous().cpu().numpy(), encoder_hidden
        return lat, encoder_hidden

    # ---------------------------------------------------------------------------- learning
    def learn(self, batch, t_env, keep=None):
        """nova/behavior_FC_policy.py:142-228 for all agents and all windows at once."""
        a, dev = self.args, self.device
        history = batch["history"][:, :-1].to(device=dev, dtype=torch.float32)      # [E, T, nA, N, d]
        E, T, nA, N, d = history.shape
        L, Z = self.max_history_len, self.latent_dim
        J = T - 1 - L
        # right-aligned, zero-padded windows (behavior_traj_wrapper :110-138): curr_j = steps j-L+1..j, next_j = j-L+2..j+1
        pad = torch.cat([torch.zeros(E, L - 1, nA, N, d, device=dev), history], dim=1)
        win = pad.unfold(1, L, 1)                                                    # [E, T, nA, N, d, L] (window k = steps k-L+1..k)
        win = win.permute(2, 0, 3, 1, 5, 4)                                          # [nA, E, N, T, L, d]
        curr = win[:, :, :, :J].reshape(nA, E * N * J, L * d).contiguous()
        nxt = win[:, :, :, 1:J + 1].reshape(nA, E * N * J, L * d).contiguous()
        rows = E * N * J
        enc = ops.mlp3_forward(self.enc_arena, "", curr, a.encoder_rnn_dim, Z, softmax=True)
        lat = enc["out"].reshape(nA, E * N, J, Z)
        lat_prev = torch.cat([torch.zeros(nA, E * N, 1, Z, device=dev), lat[:, :, :-1]], dim=2).reshape(nA, rows, Z)
        dec_in = torch.cat([curr, lat_prev], dim=-1).contiguous()
        dec = ops.mlp3_forward(self.dec_arena, "decoder.", dec_in, a.decoder_rnn_dim, L * d, target=nxt)
        # loss_j = sum|next - pred| / (E N L d + EPS) * d * N, averaged over the J windows; data-parallel runs count the
        # envs of all ranks
        dp = getattr(self, "dp", None)
        cnt = float(E * N * L * d) * (dp.world if dp is not None else 1)
        scale = float(d * N) / (cnt + EPS) / J
        loss = dec["l1"] * scale                                                     # [nA]
        dx = ops.mlp3_backward(self.dec_arena, "decoder.", dec, g_scale=scale, want_dx=True)
        g_lat_prev = dx[:, :, L * d:].reshape(nA, E * N, J, Z)
        g_lat = torch.cat([g_lat_prev[:, :, 1:], torch.zeros(nA, E * N, 1, Z, device=dev)], dim=2).reshape(nA, rows, Z).contiguous()
        ops.mlp3_backward(self.enc_arena, "", enc, g_out=g_lat)
        if dp is not None:
            dp.all_reduce_grads(self.enc_arena, self.dec_arena)
        sq = step_all(self.behavior_optimizer, self.max_grad_norm if self._use_max_grad_norm else None)
        host = torch.cat([loss, sq.sqrt().reshape(-1)]).cpu()                        # ONE host read-back
        losses, norms = host[:nA].numpy(), host[nA:].reshape(nA, 2)
        train_info = {"behavior_loss": float(losses.sum()), "stability_loss": 0.0, "behavior_total": float(losses.sum()),
                      "behavior_encoder_grad_norm": float(norms[:, 0].sum()), "behavior_decoder_grad_norm": float(norms[:, 1].sum())}
        if t_env - self.log_stats_t >= self.args.learner_log_interval:
            for k, v in train_info.items():
                self.logger.log_stat(self.log_prefix + k, v, t_env)
        beh = [np.asarray(losses[i]) for i in range(nA)]
        return beh, [], [np.asarray(losses[i]) for i in range(nA)]
