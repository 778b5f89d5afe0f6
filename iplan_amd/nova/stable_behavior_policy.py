"""Behavior_policy (soft update = iPLAN) -- behavioural-incentive inference module (mirror of
nova/stable_behavior_policy.py:13-312)."""
import copy
import contextlib
import os

import numpy as np
import torch

from .. import ops
from ..arena import ParamArena
from ..optim import FusedAdam, step_all
from ..streams import AsyncHost, distinct_stream, masked_stream, probe_mode
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
    def latent_update(self, history, encoder_hidden, prev_latent, out_latent=None, out_hidden=None, launch=True):
        """history [E,nA,N,L,d], encoder_hidden [E,layers,nA,N,R], prev_latent [E,nA,N,Z] ->
        (new_latent [E,nA,N,Z], new_hidden [E,layers,nA,N,R])  (stable_behavior_policy.py:83-123).
        numpy history -> numpy latent + torch hidden (as the reference returns); device tensors in
        -> device tensors out.  One fused launch for all agents, soft update included.  ``history`` may be any
        strided device view (a sliding window over a time-major log is read in place); ``out_latent`` [E,nA,N,Z]
        / ``out_hidden`` [E,nA,N,R]: optional destination views written by the kernel."""
        as_np = isinstance(history, np.ndarray)
        hist = _as_dev(history, self.device)
        hid = _as_dev(encoder_hidden, self.device)
        prev = _as_dev(prev_latent, self.device)
        E, nA, N, L, d = hist.shape
        res = ops.enc_forward(self.enc_arena, hist.permute(1, 0, 2, 3, 4), hid[:, 0].permute(1, 0, 2, 3),
                              prev.permute(1, 0, 2, 3), self.soft_update_coef, self.latent_dim,
                              out_lat=None if out_latent is None else out_latent.permute(1, 0, 2, 3),
                              out_h=None if out_hidden is None else out_hidden.permute(1, 0, 2, 3), launch=launch)
        if not launch:                                    # device-resident rollout: the launch is fused into the GAT's
            return res[2]                                 # (``Prediction_policy.GAT_latent_update(fuse_enc=...)``)
        lat, hL = res
        lat = lat.permute(1, 0, 2, 3)                     # [E, nA, N, Z]
        hL = hL.permute(1, 0, 2, 3).unsqueeze(1)          # [E, 1, nA, N, R]
        if as_np:
            return lat.cpu().numpy(), hL
        return lat, hL


    # ---------------------------------------------------------------------------- learning
    def behavior_traj_wrapper(self, history, step, mask):
        """nova/stable_behavior_policy.py:128-157, vectorised: window ``step`` of one agent's episode.
        history [E, T, N, d], mask [E, T] -> (curr [E,N,L,d] = steps step-L+1..step right-aligned / zero-padded,
        next [E,N,L,d] = steps step+1..step+L, and their per-step masks broadcast to the same shapes; padded positions of the
        current mask stay 1 like the reference's).  ``learn`` does not call this -- the kernels walk the windows in place --
        it is kept for callers that want the reference's per-window tensors."""
        history, mask = torch.as_tensor(history), torch.as_tensor(mask)
        E, T, N, d = history.shape
        L = self.max_history_len
        start, plug = max(0, step - L + 1), max(0, L - step - 1)
        curr = torch.zeros(E, N, L, d, dtype=history.dtype, device=history.device)
        curr[:, :, plug:] = history[:, start:step + 1].permute(0, 2, 1, 3)
        nxt = history[:, step + 1:step + L + 1].permute(0, 2, 1, 3)
        m = mask.to(history.dtype)
        m_curr = torch.ones_like(curr)
        m_curr[:, :, plug:] = m[:, start:step + 1, None, None].permute(0, 2, 1, 3).expand(E, N, step + 1 - start, d)
        m_next = m[:, step + 1:step + L + 1, None, None].permute(0, 2, 1, 3).expand(E, N, L, d).contiguous()
        return curr, nxt, m_curr, m_next

    def _global_window_sums(self, mask, hard=False):
        """The loss normalisers (mask sums per window), over ALL ranks' envs in data-parallel runs.  Always handed to the
        kernels: their in-kernel fallback re-sums E x L mask entries per window in every wave (a dependent-load loop that
        cost the decoder kernels about a tenth of their time at 110 envs)."""
        dp = getattr(self, "dp", None)
        if dp is None and os.environ.get("IPLAN_BEH_KERNEL_WINSUM"):             # (timing experiments: the in-kernel sums)
            return None
        wn = ops.beh_window_mask_sums(mask, self.max_history_len, hard=hard)
        return wn if dp is None else dp.all_reduce_sum(wn)

    def _global_envs(self, E):
        """envs the stability statistic is averaged over: this process's, times the data-parallel world size"""
        dp = getattr(self, "dp", None)
        return E * (dp.world if dp is not None else 1)

    def join_decoder(self):
        """Make the current stream wait for a decoder update that ``learn(..., defer_decoder=True)`` left running on the side
        stream (no-op otherwise).  Everything that reads or writes the decoder's parameters calls this first."""
        self.flush_decoder()
        ev = getattr(self, "_dec_done", None)
        if ev is not None and self.defer_late == 1:
            # (the update was enqueued moments ago: a host wait here would stall the enqueue of this call's own forward; the
            # allocator alternates between two sets of record blocks instead)
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._dec_done = None
        elif ev is not None:
            # HOST wait, not just a stream wait: the update holds the previous call's 23 GB of BPTT records (record_stream);
            # until its event has completed the caching allocator cannot hand those blocks to this call and would go to
            # hipMalloc for fresh ones (+13 ms per call, measured)
            ev.synchronize()
            torch.cuda.current_stream(self.device).wait_event(ev)
            self._dec_done = None

    defer_late = int(os.environ.get("IPLAN_DEFER_LATE", "0"))

    def flush_decoder(self):
        """Enqueue a decoder update that ``learn(..., defer_decoder=True)`` held back (IPLAN_DEFER_LATE), behind the work the
        current stream holds now.  The training loop calls it when the next rollout has been enqueued; ``join_decoder`` calls it too."""
        f = getattr(self, "_dec_pending", None)
        if f is not None:
            self._dec_pending = None
            f()

    learn_takes_prepared = True         # (subclasses with their own ``learn`` signature switch it off)

    def prepare_learn(self, batch):
        """The data-movement head of ``learn``: episode views, the [nA, E, T] float mask (polarity as the reference's, :190-193) and the
        per-window mask sums (all ranks' in data-parallel runs).  A training loop that enqueues other learners beside ``learn`` calls
        this FIRST and passes the result as ``learn(..., prepared=)``: these few tiny launches otherwise queue behind the other
        learners' kernels at the head of the learn phase."""
        a, dev = self.args, self.device
        history = batch["history"][:, :-1].to(device=dev, dtype=torch.float32)     # [E, T, nA, N, d]
        term = batch["terminated"][:, :-1].to(dev)                                 # [E, T, nA, 1]
        mask = (1 - term[..., 0]) if a.env == "MPE" else term[..., 0]
        mask = mask.permute(2, 0, 1).to(torch.float32).contiguous()                # [nA, E, T]
        return dict(batch=batch, history=history, mask=mask, hist=history.permute(2, 0, 1, 3, 4),   # hist: [nA, E, T, N, d] view
                    win_norm=self._global_window_sums(mask))

    def learn(self, batch, t_env, keep=None, defer_decoder=False, defer_readback=False, prepared=None):
        """nova/stable_behavior_policy.py:161-279 for all agents at once: ONE persistent forward launch
        walks every (env, entity) chain through the T-1-L windows (decoder + encoder GRUs, soft latent
        update, masked L1), ONE backward launch does the BPTT, then weight-gradient contractions,
        separate clipping of the encoder and decoder groups and Adam.  ``keep`` (uint8 dropout keep flags
        [nA, J, E*N, L, 64]) may be injected; by default the kernel draws them from a counter-based
        generator seeded from torch's RNG.  Returns (behavior_loss, stability_loss, total_loss) lists.

        ``defer_decoder=True`` (device-resident training loops): the DECODER's weight-gradient contraction, gradient
        all-reduce, clip and Adam step are enqueued on a side stream and this call returns once the encoder is updated.
        Nothing outside behaviour learning reads the decoder (the rollout uses the encoder only), so that work -- the largest
        bandwidth-bound item of a ``learn`` -- runs beside the next rollout, whose kernels leave 96 of the 256 CUs idle; the
        next ``learn`` (and save / load) waits for it.  Same arithmetic, same order of optimiser steps; the logged decoder
        gradient norm is the previous call's.

        ``defer_readback=True``: the call does not wait for the device at all -- the losses / gradient norms are staged to
        pinned host memory behind the enqueued work and the call returns a function that delivers the three lists (and logs)
        when called; the host can then enqueue the next rollout while this call's kernels still run."""
        a = self.args
        dev = self.device
        self.join_decoder()
        if prepared is None or prepared["batch"] is not batch:
            prepared = self.prepare_learn(batch)
        history, mask, hist, wn_all = prepared["history"], prepared["mask"], prepared["hist"], prepared["win_norm"]
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if keep is None else 0
        E, N = history.shape[0], self.max_vehicle_num
        chunk = int(getattr(a, "behavior_env_chunk", 0) or os.environ.get("IPLAN_BEH_ENV_CHUNK", "128"))
        if E <= chunk:
            fwd = ops.beh_forward(self.enc_arena, self.dec_arena, hist, mask, self.max_history_len, self.latent_dim,
                                  self.soft_update_coef, self.thres_small_variation, a.decoder_dropout, keep=keep, seed=seed,
                                  win_norm=wn_all)
            defer = bool(defer_decoder)
            bwd = ops.beh_backward(self.enc_arena, self.dec_arena, fwd, penalty=self.behavior_variation_penalty,
                                   E_norm=self._global_envs(E), defer_dec_wgrad=defer)
            loss_dev = fwd["loss"]
        else:
            defer = False
            if defer_decoder and not getattr(self, "_warned_chunked_defer", False):
                self._warned_chunked_defer = True
                print(f"Behavior_policy.learn: defer_decoder ignored -- {E} envs are learnt in chunks of {chunk} "
                      "(one optimiser step behind the last chunk, decoder included)")
            # Large batches (config 4's 256 envs on one GPU: 230 GB of BPTT records at once): env chunks run one after the
            # other -- chains never interact; the loss normalisers are the window mask sums over ALL envs (and ranks), so
            # the chunk gradients simply add up in the arenas -- and one clip + Adam step follows.
            wn = wn_all
            loss_dev = None
            for c, lo in enumerate(range(0, E, chunk)):
                hi = min(E, lo + chunk)
                fwd = ops.beh_forward(self.enc_arena, self.dec_arena, hist[:, lo:hi], mask[:, lo:hi].contiguous(), self.max_history_len,
                                      self.latent_dim, self.soft_update_coef, self.thres_small_variation, a.decoder_dropout,
                                      keep=None if keep is None else keep[:, :, lo * N:hi * N].contiguous(),
                                      seed=seed + 0x9E3779B97F4A7C15 * c & (2 ** 63 - 1), win_norm=wn)
                ops.beh_backward(self.enc_arena, self.dec_arena, fwd, accumulate=c > 0, penalty=self.behavior_variation_penalty,
                                 E_norm=self._global_envs(E))
                part = fwd["loss"] * torch.tensor([1.0, (hi - lo) / E], device=dev)      # the stability statistic is a mean over envs
                loss_dev = part if loss_dev is None else loss_dev + part
                del fwd
        max_norm = self.max_grad_norm if self._use_max_grad_norm else None
        nA = self.n_agents
        if E <= chunk and defer:
            # encoder now (the next rollout needs it) ...
            if getattr(self, "dp", None) is not None:
                self.dp.all_reduce_grads(self.enc_arena)
            sq = step_all(self.behavior_optimizer, max_norm, slices=(0,))
            steps = self.behavior_optimizer[0].last_steps
            prev_dec = getattr(self, "_dec_sq", None)
            # ... decoder on the side stream, same optimiser step
            on_gpu = torch.device(dev).type == "cuda"
            if on_gpu and getattr(self, "_dec_stream", None) is None:
                # plain side stream by default; IPLAN_DEFER_CUS=k restricts it to k CUs (see harness.py / profiles/r02e_notes.md)
                cus = int(os.environ.get("IPLAN_DEFER_CUS", "0"))
                # (beside the NEXT rollout: not on the caller's hardware queue, streams.distinct_stream)
                # (beside the NEXT rollout: not on the caller's hardware queue when queues are probed, streams.distinct_stream.  Sharing the
                # prediction learner's stream instead was measured: no better without RCCL, worse with it -- profiles/r06_notes.md section 8)
                self._dec_stream = masked_stream(dev, cus) if cus > 0 else (
                    distinct_stream(dev, [torch.cuda.current_stream(dev)]) if probe_mode() == "full" else torch.cuda.Stream(dev))
            ds = self._dec_stream if on_gpu else None                   # (CPU / emulator: same code, run in line)

            def update(bwd=bwd):
                bwd["dec_wgrad"](ds)                                    # (behind whatever the CALLER's stream holds at this point)
                with (torch.cuda.stream(ds) if on_gpu else contextlib.nullcontext()):
                    if getattr(self, "dp", None) is not None:
                        self.dp.all_reduce_grads(self.dec_arena)
                    # (its own norm buffer: the encoder half of the NEXT call fills the shared one on the main stream while this
                    # half may still be running here)
                    if getattr(self, "_dec_sq_buf", None) is None:
                        self._dec_sq_buf = torch.zeros(nA, len(self.behavior_optimizer[0].slices), dtype=torch.float32, device=dev)
                    sq_d = step_all(self.behavior_optimizer, max_norm, slices=(1,), steps=steps, sq=self._dec_sq_buf)
                    self._dec_sq = sq_d[:, 1].clone()
                    if on_gpu:
                        self._dec_done = torch.cuda.Event()
                        self._dec_done.record(ds)

            # IPLAN_DEFER_LATE=1 (A/B knob): the update is not enqueued here but handed to the training loop, which starts it
            # (flush_decoder) once the NEXT rollout is on the device -- the wide contraction holds every SIMD for ~2.7 ms, and
            # started here it does so in front of that rollout's first, latency-bound launches.  Same values either way.
            if on_gpu and self.defer_late:
                self._dec_pending = update
            else:
                update()
            del update
            del bwd
            dec_col = prev_dec if prev_dec is not None else torch.zeros(nA, device=dev)
            host_dev, split = torch.cat([loss_dev.reshape(-1), sq[:, 0].sqrt(), dec_col.sqrt()]), True
        else:
            if getattr(self, "dp", None) is not None:
                self.dp.all_reduce_grads(self.enc_arena, self.dec_arena)
            sq = step_all(self.behavior_optimizer, max_norm)
            host_dev, split = torch.cat([loss_dev.reshape(-1), sq.sqrt().reshape(-1)]), False
        if torch.device(dev).type == "cuda" and not getattr(self, "_queues_checked", False):
            # first learn() done: the encoder's side streams have their hardware queues -- not the caller's, or they get replaced
            # (ops.verify_side_queues; a training loop with more streams to place, harness.cycle, runs its own wider check)
            self._queues_checked = True
            if not defer_readback:
                ops.verify_side_queues(dev, torch.cuda.current_stream(dev))
        staged = AsyncHost(host_dev) if defer_readback else None

        def finish():
            host = staged.get() if staged is not None else host_dev.cpu()              # ONE host read-back
            loss = host[:2 * nA].reshape(nA, 2).numpy()
            norms = host[2 * nA:].reshape(2, nA).t() if split else host[2 * nA:].reshape(nA, 2)
            beh = [np.asarray(loss[i, 0]) for i in range(nA)]
            stab = [np.asarray(loss[i, 1]) for i in range(nA)]
            total = [np.asarray(loss[i, 0] + self.behavior_variation_penalty * loss[i, 1]) for i in range(nA)]
            train_info = {"behavior_loss": float(loss[:, 0].sum()), "stability_loss": float(loss[:, 1].sum()),
                          "behavior_total": float(sum(float(t) for t in total)),
                          "behavior_encoder_grad_norm": float(norms[:, 0].sum()),
                          # deferred decoder update: the norm of the PREVIOUS call's decoder step (this call's is still on its
                          # way; 0 on the first call) -- flagged by the extra stat below
                          "behavior_decoder_grad_norm": float(norms[:, 1].sum())}
            if split:
                train_info["behavior_decoder_grad_norm_lag_calls"] = 1.0
            if t_env - self.log_stats_t >= self.args.learner_log_interval:
                for k, v in train_info.items():
                    self.logger.log_stat(self.log_prefix + k, v, t_env)
            return beh, stab, total
        return finish if defer_readback else finish()

    # ---------------------------------------------------------------------------- checkpoints
    def save_models(self, path):
        self.join_decoder()
        for i in range(self.n_agents):
            torch.save(self.behavior_encoder[i].state_dict(), f"{path}/behavior_encoder_{i}.th")
            torch.save(self.behavior_decoder[i].state_dict(), f"{path}/behavior_decoder_{i}.th")
            torch.save(self.behavior_optimizer[i].state_dict(), f"{path}/behavior_optimizer_{i}_opt.th")

    def load_models(self, paths, load_optimisers=False):
        self.join_decoder()
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
