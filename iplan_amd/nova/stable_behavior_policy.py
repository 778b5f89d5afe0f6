"""Behavior_policy (soft update = iPLAN) -- behavioural-incentive inference module (mirror of
nova/stable_behavior_policy.py:13-312)."""
import copy

import numpy as np
import torch

from .. import ops
from ..arena import ParamArena
from ..optim import FusedAdam
from .behavior_net import Behavior_Latent_Decoder, EncoderRNN
from .prediction_policy import _as_dev

EPS = 1e-10


class Behavior_policy:
    def __init__(self, args, logger):
        self.device = torch.device("cuda" if args.use_cuda else "cpu")
        self.args = args
        self.n_actions = args.n_actions
        self.n_agents = args.n_agents
        self.max_vehicle_num = args.max_vehicle_num
        self.max_history_len = args.max_history_len
        self.latent_dim = args.latent_dim
        self.optim_eps = args.optim_eps
        self.weight_decay = args.weight_decay
        self.obs_shape = args.obs_shape
        self.init_behavior_net()
        self.logger = logger
        self.log_prefix = args.log_prefix
        self.log_stats_t = -self.args.learner_log_interval - 1
        self._use_max_grad_norm = args.use_max_grad_norm
        self.max_grad_norm = args.max_grad_norm
        self.soft_update_coef = args.soft_update_coef
        self.behavior_variation_penalty = args.behavior_variation_penalty
        self.thres_small_variation = args.thres_small_variation

    def init_behavior_net(self):
        """nova/stable_behavior_policy.py:56-80."""
        a = self.args
        self.behavior_encoder, self.behavior_decoder = [], []
        for _ in range(self.n_agents):
            self.behavior_encoder.append(EncoderRNN(input_size=a.obs_shape_single, hidden_size=a.encoder_rnn_dim,
                                                    output_size=a.latent_dim, num_layers=a.num_encoder_layer))
            self.behavior_decoder.append(Behavior_Latent_Decoder(
                input_size=a.obs_shape_single + a.latent_dim, hidden_size=a.decoder_rnn_dim,
                output_size=a.obs_shape_single, num_layers=a.num_decoder_layer, dropout=a.decoder_dropout))
        self.enc_arena = ParamArena(self.behavior_encoder, self.device)
        self.dec_arena = ParamArena(self.behavior_decoder, self.device)
        for i in range(self.n_agents):
            self.behavior_encoder[i].attach(self.enc_arena, i)
            self.behavior_decoder[i].attach(self.dec_arena, i)
        self.behavior_optimizer = [
            FusedAdam([(self.enc_arena, i), (self.dec_arena, i)], lr=a.lr_behavior, eps=self.optim_eps,
                      weight_decay=self.weight_decay) for i in range(self.n_agents)]

    # ---------------------------------------------------------------------------- rollout
    def latent_update(self, history, encoder_hidden, prev_latent):
        """history [E,nA,N,L,d], encoder_hidden [E,layers,nA,N,R], prev_latent [E,nA,N,Z] ->
        (new_latent [E,nA,N,Z], new_hidden [E,layers,nA,N,R])  (stable_behavior_policy.py:83-123).
        numpy history -> numpy latent + torch hidden (as the reference returns); device tensors in
        -> device tensors out.  One fused launch for all agents, soft update included."""
        as_np = isinstance(history, np.ndarray)
        hist = _as_dev(history, self.device)
        hid = _as_dev(encoder_hidden, self.device)
        prev = _as_dev(prev_latent, self.device)
        E, nA, N, L, d = hist.shape
        lat, hL = ops.enc_forward(self.enc_arena, hist.permute(1, 0, 2, 3, 4), hid[:, 0].permute(1, 0, 2, 3),
                                  prev.permute(1, 0, 2, 3), self.soft_update_coef, self.latent_dim)
        lat = lat.permute(1, 0, 2, 3)                     # [E, nA, N, Z]
        hL = hL.permute(1, 0, 2, 3).unsqueeze(1)          # [E, 1, nA, N, R]
        if as_np:
            return lat.cpu().numpy(), hL
        return lat, hL

    # ---------------------------------------------------------------------------- checkpoints
    def save_models(self, path):
        for i in range(self.n_agents):
            torch.save(self.behavior_encoder[i].state_dict(), f"{path}/behavior_encoder_{i}.th")
            torch.save(self.behavior_decoder[i].state_dict(), f"{path}/behavior_decoder_{i}.th")
            torch.save(self.behavior_optimizer[i].state_dict(), f"{path}/behavior_optimizer_{i}_opt.th")

    def load_models(self, paths, load_optimisers=False):
        if len(paths) == 1:
            paths = [copy.copy(paths[0]) for _ in range(self.n_agents)]
        for i in range(self.n_agents):
            self.behavior_encoder[i].load_state_dict(
                torch.load(f"{paths[i]}/behavior_encoder_{i}.th", map_location="cpu"))
            self.behavior_decoder[i].load_state_dict(
                torch.load(f"{paths[i]}/behavior_decoder_{i}.th", map_location="cpu"))
            if load_optimisers:
                self.behavior_optimizer[i].load_state_dict(
                    torch.load(f"{paths[i]}/behavior_optimizer_{i}_opt.th", map_location="cpu"))
