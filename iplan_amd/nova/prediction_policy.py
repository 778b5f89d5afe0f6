"""Prediction_policy -- instant-incentive inference module (mirror of nova/prediction_policy.py:14-286).

Owns n_agents x (GAT_Net, Prediction_Decoder) with one optimiser per agent, exactly like the
reference, but the weights of all agents live in two stacked arenas so that
``GAT_latent_update`` is ONE fused launch over (agent, env) and ``learn`` is one forward and one
backward launch sequence for all agents.
"""
import copy

import numpy as np
import torch

from .. import ops
from ..arena import ParamArena
from ..optim import FusedAdam, step_all
from ..streams import AsyncHost
from .GAT_Net import GAT_Net, gumbel_noise
from .prediction_net import Prediction_Decoder

EPS = 1e-10


def _as_dev(x, device):
    """numpy / tensor -> fp32 tensor on device (the runner hands float64 numpy arrays over)."""
    if isinstance(x, np.ndarray):
        return torch.as_tensor(x, dtype=torch.float32).to(device)
    return x.to(device=device, dtype=torch.float32)


class Prediction_policy:
    def __init__(self, args, logger):
        self.device = torch.device("cuda" if args.use_cuda else "cpu")
        self.args = args
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        self.max_vehicle_num = args.max_vehicle_num
        self.max_history_len = args.max_history_len
        self.max_episode_len = args.episode_limit
        self.prediction_batch_size = args.pred_batch_size
        self.pred_length = args.pred_length
        self.optim_eps = args.optim_eps
        self.weight_decay = args.weight_decay
        self.obs_shape = args.obs_shape_single
        self.logger = logger
        self.log_prefix = args.log_prefix
        self.log_stats_t = -self.args.learner_log_interval - 1
        self.GAT_input_dim = args.obs_shape_single + (args.latent_dim if args.GAT_use_behavior else 0)
        self.init_GAT_net()
        self._use_max_grad_norm = args.use_max_grad_norm
        self.max_grad_norm = args.max_grad_norm

    def init_GAT_net(self):
        """nova/prediction_policy.py:64-88 (same construction order -> same seeded init)."""
        a = self.args
        self.pred_GAT, self.pred_decoder = [], []
        for _ in range(self.n_agents):
            self.pred_GAT.append(GAT_Net(input_shape=self.GAT_input_dim, args=a))
            self.pred_decoder.append(Prediction_Decoder(
                input_size=a.obs_shape_single, hidden_size=a.attention_dim, output_size=a.obs_shape_single,
                num_layers=1, pred_length=a.pred_length, teacher_forcing_ratio=a.teacher_forcing_ratio,
                dropout=a.decoder_dropout))
        self.gat_arena = ParamArena(self.pred_GAT, self.device)
        self.dec_arena = ParamArena(self.pred_decoder, self.device)
        ParamArena.colocate_grads([self.gat_arena, self.dec_arena])      # one gradient collective per learn() instead of two
        for i in range(self.n_agents):
            self.pred_GAT[i].attach(self.gat_arena, i)
            self.pred_decoder[i].attach(self.dec_arena, i)
        self.pred_optimizer = [
            FusedAdam([(self.gat_arena, i), (self.dec_arena, i)], lr=a.lr_predict, eps=self.optim_eps,
                      weight_decay=self.weight_decay) for i in range(self.n_agents)]

    # ---------------------------------------------------------------------------- rollout
    def GAT_latent_update(self, history_single, encoder_hidden, behavior_latent=None, noise=None, out=None, fuse_enc=None, fuse_ac=None):
        """history_single [E,nA,N,d], encoder_hidden [E,nA,N,A], behavior_latent [E,nA,N,Z] ->
        attention latent [E,nA,N,A]  (nova/prediction_policy.py:92-118).
        numpy in -> numpy out (drop-in for ParallelRunner); device tensors in -> device tensor out.
        ``noise``: pre-drawn gumbel samples [nA,E,N,N-1,2]; ``out``: optional [E,nA,N,A] destination view
        (e.g. ``batch["attention_latent"][:, t + 1]``) the kernel writes in place."""
        as_np = isinstance(history_single, np.ndarray)
        hist = _as_dev(history_single, self.device)
        hid = _as_dev(encoder_hidden, self.device)
        E, nA, N, _ = hist.shape
        lat = None
        if self.args.GAT_use_behavior:
            lat = _as_dev(behavior_latent, self.device).permute(1, 0, 2, 3)
        if noise is None:
            noise = gumbel_noise((nA, E, N, N - 1, 2), self.device)
        out, _ = ops.gat_forward(self.gat_arena, hist.permute(1, 0, 2, 3), lat, hid.permute(1, 0, 2, 3), noise,
                                 out=None if out is None else out.permute(1, 0, 2, 3), fuse_enc=fuse_enc, fuse_ac=fuse_ac)
        out = out.permute(1, 0, 2, 3)          # [E, nA, N, A] view
        return out.cpu().numpy() if as_np else out


    # ---------------------------------------------------------------------------- learning
    def _sample(self, n_thread, avail_len):
        """The host-side random draws of one agent in the reference's order: the (episode, t) sample of
        prediction_batch_wrapper (nova/prediction_policy.py:146) and the per-step teacher-forcing coin
        of Prediction_Decoder.forward (nova/prediction_net.py:56; drawn even at ratio 0)."""
        sel = np.random.choice(n_thread * avail_len, size=self.prediction_batch_size, replace=False)
        coins = [np.random.random() < self.args.teacher_forcing_ratio for _ in range(self.pred_length)]
        return sel, coins

    def prediction_batch_wrapper(self, history, attention_rnn, mask, behavior_latent=None):
        """nova/prediction_policy.py:122-164, vectorised: sample ``pred_batch_size`` (episode, t) pairs of one agent's
        episodes (same ``np.random.choice`` draw as the reference) and gather the input state, the stored attention /
        behaviour latents, the next ``pred_length`` states and the mask.  history [E,T,N,d], attention_rnn [E,T,N,A],
        mask [E,T], behavior_latent [E,T,N,Z] -> (input_traj [S,N,1,d], input_attention [S,N,1,A], input_latent [S,N,1,Z] or
        None, actual_traj [S,N,P,d], mask_over_traj [S,N,P,d]).  ``learn`` gathers for all agents at once instead."""
        history, attention_rnn, mask = torch.as_tensor(history), torch.as_tensor(attention_rnn), torch.as_tensor(mask)
        E, T, N, d = history.shape
        S, P = self.prediction_batch_size, self.pred_length
        avail_len = T - P - 1
        sel = torch.as_tensor(np.random.choice(E * avail_len, size=S, replace=False), dtype=torch.long, device=history.device)
        bi, ti = sel // avail_len, sel % avail_len
        input_traj = history[bi, ti].unsqueeze(2)
        input_attention = attention_rnn[bi, ti].unsqueeze(2)
        input_latent = torch.as_tensor(behavior_latent)[bi, ti].unsqueeze(2) if self.args.GAT_use_behavior else None
        steps = ti[:, None] + 1 + torch.arange(P, device=history.device)[None, :]
        actual_traj = history[bi[:, None], steps].permute(0, 2, 1, 3)
        mask_over_traj = mask[bi, ti].to(history.dtype)[:, None, None, None].expand(S, N, P, d).contiguous()
        return input_traj, input_attention, input_latent, actual_traj, mask_over_traj

    def learn(self, batch, t_env, noise=None, keep=None, defer=False, sel=None):
        """nova/prediction_policy.py:168-253 for all agents at once: sample -> fused GAT forward ->
        decoder forward + masked L1 -> decoder backward -> GAT backward -> weight gradients ->
        separate clip of the GAT and decoder groups -> Adam.  ``noise`` (gumbel, [nA, S, N, N-1, 2]) and
        ``keep`` (dropout keep flags, [nA, P, S*N, A]) may be injected; by default they are drawn from
        torch's generator on the device.  Returns the list of n_agents losses (numpy scalars); with ``defer=True``
        everything is enqueued on the current stream and a ``finish()`` callable is returned instead (it does the
        single host read-back and the logging), so independent learners can overlap on separate streams."""
        a = self.args
        dev = self.device
        nA, N, S, P = self.n_agents, self.max_vehicle_num, self.prediction_batch_size, self.pred_length
        history = batch["history"][:, :-1].to(device=dev, dtype=torch.float32)
        attention = batch["attention_latent"][:, :-1].to(device=dev, dtype=torch.float32)
        latent = batch["behavior_latent"][:, :-1].to(device=dev, dtype=torch.float32)
        term = batch["terminated"][:, :-1].to(dev)
        E, T = history.shape[:2]
        d = history.shape[-1]
        avail_len = T - P - 1
        sels, coins = zip(*[self._sample(E, avail_len) for _ in range(nA)])
        if sel is None:
            sel = torch.as_tensor(np.stack(sels), dtype=torch.long, device=dev)    # [nA, S]
        else:                                                                       # injected (episode * avail_len + t) indices
            sel = torch.as_tensor(np.asarray(sel), dtype=torch.long, device=dev)
            S = sel.shape[1]
        bi, ti = sel // avail_len, sel % avail_len
        ag = torch.arange(nA, device=dev)[:, None].expand(nA, S)
        # gathers = data movement only (prediction_batch_wrapper, :123-164)
        x0 = history[bi, ti, ag].contiguous()                                      # [nA, S, N, d]
        att = attention[bi, ti, ag].contiguous()                                   # [nA, S, N, A]
        lat = latent[bi, ti, ag].contiguous() if a.GAT_use_behavior else None
        steps = ti[:, :, None] + 1 + torch.arange(P, device=dev)[None, None, :]
        actual = history[bi[:, :, None], steps, ag[:, :, None]].permute(0, 1, 3, 2, 4).contiguous()   # [nA, S, N, P, d]
        mask = term[bi, ti, ag, 0].to(torch.float32).contiguous()                  # [nA, S]  (polarity as the reference: :191)
        if noise is None:
            noise = gumbel_noise((nA, S, N, N - 1, 2), dev)
        if keep is None and a.decoder_dropout > 0:
            keep = torch.empty(nA, P, S * N, a.attention_dim, device=dev).bernoulli_(1.0 - a.decoder_dropout)
        teacher = None
        if any(any(c) for c in coins):
            teacher = torch.as_tensor(np.array(coins, dtype=np.int32), device=dev).contiguous()
        hid, saved = ops.gat_forward(self.gat_arena, x0, lat, att, noise, save=True)
        # data-parallel runs: the loss normaliser counts the samples of ALL ranks (parallel.py)
        dp = getattr(self, "dp", None)
        mask_sum = dp.all_reduce_sum(mask.sum(dim=1)) if dp is not None else None
        fwd = ops.pdec_forward(self.dec_arena, x0.reshape(nA, S * N, d), hid.reshape(nA, S * N, -1),
                               actual.reshape(nA, S * N, P, d), mask, N, keep=keep, drop_p=a.decoder_dropout, teacher=teacher,
                               mask_sum=mask_sum)
        g_h0 = ops.pdec_backward(self.dec_arena, fwd)
        ops.gat_backward(self.gat_arena, saved, g_h0.reshape(nA, S, N, -1))
        if getattr(self, "dp", None) is not None:
            self.dp.all_reduce_grads(self.gat_arena, self.dec_arena)
        sq = step_all(self.pred_optimizer, self.max_grad_norm if self._use_max_grad_norm else None)
        stats = torch.cat([fwd["loss"], sq.sqrt().reshape(-1)])
        staged = AsyncHost(stats) if defer else None             # deferred: staged behind THIS stream's work, read whenever

        def finish():
            host = staged.get() if staged is not None else stats.cpu()              # ONE host read-back
            losses = host[:nA].numpy()
            norms = host[nA:].reshape(nA, 2)
            train_info = {"prediction_loss": float(losses.sum()), "pred_encoder_grad_norm": float(norms[:, 0].sum()),
                          "pred_decoder_grad_norm": float(norms[:, 1].sum())}
            if t_env - self.log_stats_t >= self.args.learner_log_interval:
                for k, v in train_info.items():
                    self.logger.log_stat(self.log_prefix + k, v, t_env)
            return [np.asarray(x) for x in losses]
        return finish if defer else finish()

    # ---------------------------------------------------------------------------- checkpoints
    def save_models(self, path):
        for i in range(self.n_agents):
            torch.save(self.pred_GAT[i].state_dict(), f"{path}/pred_GAT_{i}.th")
            torch.save(self.pred_decoder[i].state_dict(), f"{path}/pred_decoder_{i}.th")
            torch.save(self.pred_optimizer[i].state_dict(), f"{path}/pred_optimizer_{i}_opt.th")

    def load_models(self, paths, load_optimisers=False):
        if len(paths) == 1:
            paths = [copy.copy(paths[0]) for _ in range(self.n_agents)]
        for i in range(self.n_agents):
            self.pred_GAT[i].load_state_dict(torch.load(f"{paths[i]}/pred_GAT_{i}.th", map_location="cpu"))
            self.pred_decoder[i].load_state_dict(torch.load(f"{paths[i]}/pred_decoder_{i}.th", map_location="cpu"))
            if load_optimisers:
                self.pred_optimizer[i].load_state_dict(
                    torch.load(f"{paths[i]}/pred_optimizer_{i}_opt.th", map_location="cpu"))
