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
from ..optim import FusedAdam
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
        for i in range(self.n_agents):
            self.pred_GAT[i].attach(self.gat_arena, i)
            self.pred_decoder[i].attach(self.dec_arena, i)
        self.pred_optimizer = [
            FusedAdam([(self.gat_arena, i), (self.dec_arena, i)], lr=a.lr_predict, eps=self.optim_eps,
                      weight_decay=self.weight_decay) for i in range(self.n_agents)]

    # ---------------------------------------------------------------------------- rollout
    def GAT_latent_update(self, history_single, encoder_hidden, behavior_latent=None, noise=None):
        """history_single [E,nA,N,d], encoder_hidden [E,nA,N,A], behavior_latent [E,nA,N,Z] ->
        attention latent [E,nA,N,A]  (nova/prediction_policy.py:92-118).
        numpy in -> numpy out (drop-in for ParallelRunner); device tensors in -> device tensor out."""
        as_np = isinstance(history_single, np.ndarray)
        hist = _as_dev(history_single, self.device)
        hid = _as_dev(encoder_hidden, self.device)
        E, nA, N, _ = hist.shape
        lat = None
        if self.args.GAT_use_behavior:
            lat = _as_dev(behavior_latent, self.device).permute(1, 0, 2, 3)
        if noise is None:
            noise = gumbel_noise((nA, E, N, N - 1, 2), self.device)
        out, _ = ops.gat_forward(self.gat_arena, hist.permute(1, 0, 2, 3), lat, hid.permute(1, 0, 2, 3), noise)
        out = out.permute(1, 0, 2, 3)          # [E, nA, N, A] view
        return out.cpu().numpy() if as_np else out

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
