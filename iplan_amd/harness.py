"""Synthetic training-loop harness: plays the role of run_ippo.run_sequential + ParallelRunner.run
(run_ippo.py:261-332, runners/ippo_parallel_runner.py:105-281) around the hot-path classes, with the
environment and observation_wrapper replaced by pre-generated observation tensors (the simulators
are not installable offline).  Everything stays device resident: the only thing a real
runner would need back on the host per step is the [E, nA] action array.
"""
import os

import torch

from . import synth
from .controllers.dcntrl_controller import DcntrlMAC
from .learners.ippo_learner import IPPOLearner
from .nova.prediction_policy import Prediction_policy
from .nova.stable_behavior_policy import Behavior_policy


class NullLogger:
    def __init__(self):
        self.stats = {}

    def log_stat(self, key, value, t):
        self.stats[key] = value


class SyntheticLoop:
    def __init__(self, args, E, seed=0, device="cuda", n_obs_sets=2):
        self.args, self.E, self.device = args, E, device
        self.logger = NullLogger()
        torch.manual_seed(seed)
        self.scheme = synth.make_scheme(args)
        self.mac = DcntrlMAC(self.scheme, {"agents": args.n_agents}, args)
        self.prediction = Prediction_policy(args, self.logger) if args.GAT_enable else None
        self.behavior = Behavior_policy(args, self.logger) if args.Behavior_enable else None
        self.learner = IPPOLearner(self.mac, self.scheme, self.logger, args)
        self.t_env = 0
        # Behavior_policy.learn(defer_decoder=True): decoder weight gradients + optimiser step beside the next rollout
        # (IPLAN_NO_DEFER_DECODER=1: everything inside learn(), for A/B timing)
        import inspect
        self.defer_decoder = (self.behavior is not None and torch.device(device).type == "cuda"
                              and not os.environ.get("IPLAN_NO_DEFER_DECODER")
                              and "defer_decoder" in inspect.signature(self.behavior.learn).parameters)
        # IPLAN_DEFER_CUS=k (opt-in): the deferred work on a stream masked to k CUs.  At k = 64 the cycle is as fast as with the
        # plain side stream (413 vs 414 ms) and the rollout's kernels no longer queue behind the contraction (in-situ gat_fwd
        # 189 us instead of 271 us), but the optimum is narrow (48 CUs: 492 ms, 80 CUs: 435 ms -- the update has to fit inside
        # one rollout without squeezing it), so the default is the plain side stream (profiles/r02e_notes.md).
        self._defer_cus = int(os.environ.get("IPLAN_DEFER_CUS", "0")) if not os.environ.get("IPLAN_NO_CU_MASK") else 0
        gen = torch.Generator().manual_seed(seed + 1)
        T1, nA, N = args.episode_limit + 1, args.n_agents, args.max_vehicle_num
        d, L = args.obs_shape_single, args.max_history_len
        # pre-generated "environment": per-step entity observations; the L-window is a sliding view
        self.obs_sets = []
        for _ in range(n_obs_sets):
            hist = synth.make_history(gen, (T1 + L - 1, E, nA), N, d)        # [T1+L-1, E, nA, N, d]
            hist = hist.to(device)
            reward = torch.randn(T1, E, nA, 1, generator=gen).to(device)
            terminated = (torch.rand(T1, E, nA, 1, generator=gen) < 0.1).to(torch.uint8).to(device)
            self.obs_sets.append(dict(hist=hist, reward=reward, terminated=terminated))
        self._rollouts = 0
        self._graphs, self._g_batch, self._graph_failed = {}, None, False

    def new_batch(self):
        a, E = self.args, self.E
        T1, nA, N = a.episode_limit + 1, a.n_agents, a.max_vehicle_num
        dev = self.device
        z = lambda *s, dtype=torch.float32: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        data = dict(
            history=z(E, T1, nA, N, a.obs_shape_single), behavior_latent=z(E, T1, nA, N, a.latent_dim),
            attention_latent=z(E, T1, nA, N, a.attention_dim), rnn_states_actors=z(E, T1, nA, a.rnn_hidden_dim),
            rnn_states_critics=z(E, T1, nA, a.rnn_hidden_dim), reward=z(E, T1, nA, 1),
            terminated=z(E, T1, nA, 1, dtype=torch.uint8), actions=z(E, T1, nA, 1, dtype=torch.long),
            actions_onehot=z(E, T1, nA, a.n_actions), avail_actions=torch.ones(E, T1, nA, a.n_actions, dtype=torch.int32, device=dev),
            obs=z(E, T1, nA, a.obs_shape), state=z(E, T1, a.state_shape), speed=z(E, T1, nA, 1),
            filled=torch.ones(E, T1, 1, dtype=torch.long, device=dev))
        return synth.DictBatch(data, E, T1, dev)

    @torch.no_grad()
    def rollout(self):
        """One vectorised episode of E envs x T steps (ippo_parallel_runner.py:105-281 order of calls).
        Device resident: the launches of a vector step (actor / critic, then GAT + encoder as one) read their inputs from, and write their
        outputs into, the episode-buffer tensors in place; the per-step random draws (gumbel noise of the hard
        attention, the exponential race of the action sampling) are drawn for the whole rollout in two launches.

        IPLAN_ROLLOUT_GRAPH=1 (opt-in): the whole episode -- 2 x 90 launches plus the draws -- is captured
        ONCE per observation set in a HIP graph and replayed into one static episode batch.  Measured on MI355X
        (profiles/r01g_notes.md): an isolated rollout gains 2 % (22.5 vs 23.1 ms: the gaps between the three dependent
        kernels of a vector step shrink), but inside the full training cycle graph replays cost 5-15 % (the learners'
        multi-stream pipelines start later), so eager launches stay the default.  An active ops.KernelTimers sample or a
        failed capture also falls back to eager launches."""
        idx = self._rollouts % len(self.obs_sets)
        self._rollouts += 1
        obs = self.obs_sets[idx]
        if self._graph_ok(idx):
            g = self._graphs.get(idx)
            if g is None:
                g = self._capture(idx, obs)
            if g is not None:
                g.replay()
                return self._g_batch
        # Two static episode containers used alternately (IPLAN_FRESH_BATCH=1: a new one per rollout): everything that reads a
        # rollout's container -- insert_episode_batch (copies), the three learners (joined before the next rollout) -- is done
        # with it before the rollout after next starts, and a fresh container is 14 fill kernels over 150 MB in front of every
        # rollout's first launch.
        if os.environ.get("IPLAN_FRESH_BATCH"):
            batch = self.new_batch()
        else:
            if getattr(self, "_batches", None) is None:
                self._batches = [self.new_batch(), self.new_batch()]
            batch = self._batches[self._rollouts & 1]
        self._rollout_body(obs, batch)
        return batch

    def _graph_ok(self, idx):
        import os
        from . import ops
        if torch.device(self.device).type != "cuda" or os.environ.get("IPLAN_ROLLOUT_GRAPH", "0") != "1" or self._graph_failed:
            return False
        # bench.py's in-situ kernel timing brackets single launches with events, which a graph cannot hold: while it is
        # active every 8th rollout runs eagerly (and is the one that gets timed)
        return not (ops.TIMERS is not None and (self._rollouts - 1) % 8 == 0)

    def _capture(self, idx, obs):
        dev = torch.device(self.device)
        try:
            if self._g_batch is None:
                self._g_batch = self.new_batch()
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):                       # warm-up outside the capture: allocations, lazy inits
                self._rollout_body(obs, self._g_batch)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                self._rollout_body(obs, self._g_batch)
            self._graphs[idx] = g
            return g
        except Exception as e:                               # noqa: BLE001 -- any capture problem: run eagerly from now on
            print(f"[iplan_amd] rollout graph capture failed ({type(e).__name__}: {str(e)[:120]}); using eager launches")
            self._graph_failed = True
            torch.cuda.synchronize(dev)
            return None

    def _rollout_body(self, obs, batch, noise=None, q_all=None):
        """``noise`` ([T + 1, nA, E, N, N - 1, 2] gumbel samples; entry T feeds the episode-initial GAT update) and ``q_all``
        ([T, nA, E, n_actions] exponential draws of the action race) may be injected (parity tests); by default they are
        drawn on the device."""
        a, E = self.args, self.E
        T, nA, N, L = a.episode_limit, a.n_agents, a.max_vehicle_num, a.max_history_len
        dev = self.device
        D = batch.data
        for k in ("attention_latent", "behavior_latent", "rnn_states_actors", "rnn_states_critics"):
            D[k][:, 0].zero_()                               # episode-initial state (the container may be a re-used one); every
                                                             # later step of these fields is written by the rollout's launches
        hist_all = obs["hist"]                                         # [T1 + L - 1, E, nA, N, d] time-major "environment"
        # what env.step + EpisodeBatch.update would deliver step by step, copied once
        # ... from copies of the observation set in the container's own layout [E, T, ...] (kept until the set is written
        # to): three contiguous copies per rollout instead of a 0.5 ms strided elementwise kernel on the critical path
        key = (obs["hist"]._version, obs["reward"]._version, obs["terminated"]._version)
        if obs.get("_staged_key") != key:
            obs["_staged"] = (hist_all[L - 1:L + T].permute(1, 0, 2, 3, 4).contiguous(), obs["reward"][:T].permute(1, 0, 2, 3).contiguous(),
                              obs["terminated"][:T].permute(1, 0, 2, 3).contiguous())
            obs["_staged_key"] = key
        D["history"].copy_(obs["_staged"][0])
        D["reward"][:, :T].copy_(obs["_staged"][1])
        D["terminated"][:, :T].copy_(obs["_staged"][2])
        eh = torch.zeros(2, E, 1, nA, N, a.encoder_rnn_dim, device=dev)   # ping-pong encoder hidden state
        if self.prediction is not None:
            if noise is None:
                from .nova.GAT_Net import gumbel_noise
                noise = gumbel_noise((T + 1, nA, E, N, N - 1, 2), dev)
            self.prediction.GAT_latent_update(D["history"][:, 0], D["attention_latent"][:, 0], D["behavior_latent"][:, 0],
                                              noise=noise[T], out=D["attention_latent"][:, 0])
        if q_all is None:
            q_all = torch.empty(T, nA, E, a.n_actions, device=dev).exponential_()
        # The instant-incentive (GAT) and behavioural-incentive (encoder) updates of a step are independent of each
        # other (both read the PREVIOUS latents), and the GAT launch leaves 96 of the 256 CUs idle (one scene per
        # workgroup): the encoder runs beside it -- as trailing workgroups of the same launch (below), or, with
        # IPLAN_NO_FUSE_ENC=1, on a second HIP stream joined before the next action selection.
        two_streams = torch.device(dev).type == "cuda" and self.prediction is not None and self.behavior is not None
        if two_streams:
            main = torch.cuda.current_stream(dev)
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(dev)
            side = self._side
        # One launch per step for both latent updates (iplan_gat_enc_fwd: GAT's whole-CU workgroups are dispatched ahead of the
        # encoder's, and the two cross-stream joins of a step disappear); IPLAN_NO_FUSE_ENC=1: two launches on two streams
        fuse = self.prediction is not None and self.behavior is not None and not os.environ.get("IPLAN_NO_FUSE_ENC")
        # ... and the NEXT step's action selection rides in it too (iplan_gat_enc_ac_fwd): select_actions of step t + 1 reads what
        # the latent updates of step t write and nothing sits between the two (the environment steps after the action
        # selection), so its workgroups queue behind the scenes' in the same grid -- one launch per vector step.
        # IPLAN_NO_FUSE_AC=1: two launches per step (A/B knob)
        fuse_ac = fuse and not os.environ.get("IPLAN_NO_FUSE_AC") and E <= 512      # (beyond 512 rows per net the action selection is the
                                                                                    #  streaming launch shape, not the K-split one the fusion carries)
        for t in range(T):
            if not (fuse_ac and t > 0):
                self.mac.select_actions_ippo(batch, t, test_mode=False, q_noise=q_all[t], as_numpy=False, write_back=True)
            # env.step would run here; its outputs are the pre-generated tensors
            if fuse:
                window = hist_all[t + 1:t + 1 + L].permute(1, 2, 3, 0, 4)
                enc = self.behavior.latent_update(window, eh[t & 1], D["behavior_latent"][:, t], out_latent=D["behavior_latent"][:, t + 1],
                                                  out_hidden=eh[(t + 1) & 1][:, 0], launch=False)
                nxt = None
                if fuse_ac and t + 1 < T:
                    nxt = self.mac.select_actions_ippo(batch, t + 1, test_mode=False, q_noise=q_all[t + 1], as_numpy=False, write_back=True,
                                                       launch=False)
                self.prediction.GAT_latent_update(D["history"][:, t + 1], D["attention_latent"][:, t], D["behavior_latent"][:, t],
                                                  noise=noise[t], out=D["attention_latent"][:, t + 1], fuse_enc=enc, fuse_ac=nxt)
                continue
            if two_streams:
                side.wait_stream(main)
            if self.prediction is not None:
                self.prediction.GAT_latent_update(D["history"][:, t + 1], D["attention_latent"][:, t], D["behavior_latent"][:, t],
                                                  noise=noise[t], out=D["attention_latent"][:, t + 1])
            if self.behavior is not None:
                window = hist_all[t + 1:t + 1 + L].permute(1, 2, 3, 0, 4)           # [E, nA, N, L, d] sliding view, read in place
                if two_streams:
                    with torch.cuda.stream(side):
                        self.behavior.latent_update(window, eh[t & 1], D["behavior_latent"][:, t],
                                                    out_latent=D["behavior_latent"][:, t + 1], out_hidden=eh[(t + 1) & 1][:, 0])
                    main.wait_stream(side)
                else:
                    self.behavior.latent_update(window, eh[t & 1], D["behavior_latent"][:, t],
                                                out_latent=D["behavior_latent"][:, t + 1], out_hidden=eh[(t + 1) & 1][:, 0])
        return batch

    def cycle(self):
        """One iteration of run_ippo.run_sequential's loop (run_ippo.py:261-285): rollout -> insert ->
        Behavior_policy.learn -> Prediction_policy.learn -> IPPOLearner.train (acts when the buffer is
        full).  Warm-up gates (Behavior_warmup / GAT_warmup) are treated as already passed: the timed
        workload is the steady state.  Returns the number of env transitions produced."""
        dev = torch.device(self.device)
        if dev.type == "cuda" and self.defer_decoder and self._defer_cus > 0 and not getattr(self, "_in_work", False):
            # CU-masked streams are BLOCKING streams (the extension takes no flags): they synchronise implicitly with the
            # legacy default stream.  The cycle therefore runs on a non-blocking stream of its own, so that the masked decoder
            # stream really runs beside it.
            outer = torch.cuda.current_stream(dev)
            if outer.cuda_stream == 0:
                # The default stream is joined ONCE, when the work stream is created, and never again: every operation put
                # on the legacy default stream -- even an event wait -- waits for all earlier work of the blocking (masked)
                # streams and holds back their later work, which would chain each cycle's rollout behind the previous
                # cycle's decoder update.  Callers read results after a device synchronise (bench.py does).
                if getattr(self, "_work", None) is None:
                    self._work = torch.cuda.Stream(dev)
                    self._work.wait_stream(outer)
                self._in_work = True
                try:
                    with torch.cuda.stream(self._work):
                        n = self.cycle()
                finally:
                    self._in_work = False
                return n
        # (the rollout itself is NOT masked to the complement: measured slower -- it can use the decoder update's CUs again
        # as soon as that is done; the update's long-lived 372-register waves keep other workgroups off its CUs meanwhile)
        pe = getattr(self, "phase_events", None)             # diagnostics (bench.py's projection split): events at the phase boundaries
        if pe is not None and dev.type == "cuda":
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        batch = self.rollout()
        if pe is not None and dev.type == "cuda":
            e1.record()
        self.t_env += self.E * self.args.episode_limit
        self.learner.insert_episode_batch(batch)
        # (Data-parallel runs overlap the learners too: the host issues the gradient all-reduces in the same order on
        # every rank -- prediction, PPO epochs, behaviour -- and RCCL runs them in that order on its own stream,
        # each behind the event of the stream that produced its gradients.)
        if dev.type != "cuda":
            if self.behavior is not None:
                self.behavior.learn(batch, self.t_env, **({"defer_decoder": True} if self.defer_decoder else {}))
            if self.prediction is not None:
                self.prediction.learn(batch, self.t_env)
            self.learner.train(self.t_env)
            return self.E * self.args.episode_limit
        # The three learners of a cycle touch disjoint parameter sets and only READ the episode data, and the
        # behaviour kernels occupy 138 of the 256 CUs: prediction learning and the PPO update are enqueued on two
        # side streams beside them and joined before the next rollout.  Values are identical to running them one
        # after another; only the host read-backs are deferred to the join.
        main = torch.cuda.current_stream(dev)
        if getattr(self, "_lstreams", None) is None:
            # IPLAN_PPO_PRIO=1 (A/B knob): the PPO update's stream at high priority -- its ~25 small launches per epoch are a latency
            # chain that otherwise queues for CU slots behind the behaviour kernels' long-lived workgroups
            prio = -1 if os.environ.get("IPLAN_PPO_PRIO") else 0
            # Which streams share a hardware queue is decided HERE, not by creation order (streams.py): prediction learning runs at the
            # head of the phase beside the encoder's first forward ranges and the decoder's first range -- a queue of its own; the PPO
            # update's stream must not be the main stream's (train() -> learn order is explicit, below).
            from . import ops
            from .streams import distinct_stream, probe_mode
            if probe_mode() == "full":
                sides = [ops._side_stream(dev, main.cuda_stream), ops._side_stream(dev, main.cuda_stream, 2)] if self.behavior is not None else []
                pred_s = distinct_stream(dev, [main] + sides)
                # ... and the PPO update rides on the prediction learner's stream (behind its 2 ms): the four hardware queues are taken
                # (main, the encoder's two, this one), a stream of its own would land on one of the encoder's and a rank's small
                # train() launches would queue between 0.8 ms encoder ranges (rank-of-8 step 39.1 -> 40.7 ms).  IPLAN_PPO_OWN_STREAM=1: A/B
                self._lstreams = (pred_s, distinct_stream(dev, [main], priority=prio) if os.environ.get("IPLAN_PPO_OWN_STREAM") else pred_s)
            else:                                             # "min": creation order as before; only the main stream's queue is avoided
                self._lstreams = (distinct_stream(dev, [main]), torch.cuda.Stream(dev, priority=prio))
        fins = []
        # the behaviour learner's data-movement head first: its tiny launches would otherwise sit behind the side learners' kernels
        prep = self.behavior.prepare_learn(batch) if getattr(type(self.behavior), "learn_takes_prepared", False) else None
        ev = torch.cuda.Event()
        ev.record(main)
        if self.behavior is not None and hasattr(self.behavior, "flush_decoder"):
            self.behavior.flush_decoder()                    # (a held-back decoder update of the previous cycle: behind the rollout)
        # IPLAN_BEH_FIRST=1 (A/B knob): behaviour learning -- the critical path of the phase -- is enqueued BEFORE the side
        # learners (read-back staged, so the host goes straight on to them): its first kernels then start ~0.2 ms after the
        # rollout instead of behind prediction learning's ~1.2 ms of host-side sampling + enqueue
        beh_first = self.behavior is not None and bool(os.environ.get("IPLAN_BEH_FIRST"))
        late = []                                            # host read-backs of this cycle, delivered during the next one
        beh_fin = None
        if beh_first:
            kw = {"defer_decoder": True} if self.defer_decoder else {}
            if prep is not None:
                kw["prepared"] = prep
            beh_fin = self.behavior.learn(batch, self.t_env, defer_readback=True, **kw)
        for strm, fn in zip(self._lstreams, (
                (lambda: self.prediction.learn(batch, self.t_env, defer=True)) if self.prediction is not None else None,
                lambda: self.learner.train(self.t_env, defer=True))):
            if fn is None:
                continue
            strm.wait_event(ev)
            with torch.cuda.stream(strm):
                f = fn()
                done = torch.cuda.Event()
                done.record(strm)
            fins.append((f, done))
        # Buffer-full cycles (one in eight at config 3): IPPOLearner.train and Behavior_policy.learn each fill the chip, and side by side
        # they take LONGER than one after the other (round 4, scripts/dev/host_lag.py: 270-272 ms per cycle against 265-268).  Until round 6
        # which of the two happened was an accident of HIP's stream -> hardware-queue assignment (4 queues: the PPO stream happened to
        # share one with the main stream, so behaviour learning started when train() was done) -- and a live RCCL communicator, i.e. every
        # N > 1 rank, takes queues first and flips it (+3 %: profiles/r06_notes.md section 3).  The order is now explicit: the main stream
        # waits for train() before behaviour learning is enqueued -- when train() is a chip-filling one.  A data-parallel rank's share
        # (2 880 rows per agent) is a chain of small launches that DOES make progress beside the behaviour kernels (rank-of-8 learn phase
        # 23.0 ms for 15.1 + 10.1 ms of alone-times): it stays concurrent.  IPLAN_TRAIN_ORDER = serial | concurrent overrides (A/B).
        order = os.environ.get("IPLAN_TRAIN_ORDER", "auto")
        big_train = self.args.batch_size * self.args.episode_limit >= 8192
        if (order == "serial" or (order == "auto" and big_train)) and len(fins) and fins[-1][0] is not None \
                and self.behavior is not None and not beh_first:
            main.wait_event(fins[-1][1])
        # IPLAN_RUN_AHEAD=1 (opt-in; measured 362-364 vs 358 ms per cycle with it off, profiles/r02h_notes.md)
        run_ahead = self.behavior is not None and self.defer_decoder and bool(os.environ.get("IPLAN_RUN_AHEAD"))
        if beh_fin is not None:
            late.append(beh_fin) if run_ahead else beh_fin()
        elif self.behavior is not None:
            # the decoder's weight-gradient contraction + optimiser step run on beside the next rollout (see learn())
            kw = {"defer_decoder": True} if self.defer_decoder else {}
            if prep is not None:
                kw["prepared"] = prep
            if run_ahead:
                # ... and the HOST does not wait for the device either: losses / norms are staged to pinned memory behind
                # the enqueued work (streams.AsyncHost) and read one cycle later, so the next rollout's launches are
                # already queued when the last kernel of this learn() retires
                late.append(self.behavior.learn(batch, self.t_env, defer_readback=True, **kw))
            else:
                self.behavior.learn(batch, self.t_env, **kw)
        for f, done in fins:
            main.wait_event(done)
            if f is not None:
                late.append(f) if run_ahead else f()
        for f in getattr(self, "_late", ()):                 # the previous cycle's read-backs: long complete, no waiting
            f()
        self._late = late
        if not run_ahead:                                    # (the learners' read-backs above have synchronised with the rollout's stream;
            from . import ops                                #  with host run-ahead the flag is read in finish())
            ops.check_fused_sync()
        if pe is not None:
            e2.record()
            pe.append((e0, e1, e2))
        if not getattr(self, "_queues_verified", False) and self.behavior is not None and not getattr(self, "_in_work", False):
            self._verify_queues(dev, main)
        return self.E * self.args.episode_limit

    def _verify_queues(self, dev, main):
        """IPLAN_QUEUE_PROBE=verify (the default), once, after the first cycle: the streams that must run BESIDE the main stream -- the
        encoder's forward / BPTT side streams, the prediction learner's -- are probed against its hardware queue and against each other
        (ops.verify_side_queues); one that shares is replaced.  Returns the list of replaced roles (normally empty)."""
        from . import ops
        self._queues_verified = True
        extra = {"pred": self._lstreams[0]} if getattr(self, "_lstreams", None) else {}
        replaced, roles = ops.verify_side_queues(dev, main, extra)
        if "pred" in replaced:
            self._lstreams = (roles["pred"],) + tuple(self._lstreams[1:])
        self.queue_repairs = replaced
        return replaced

    def finish(self):
        """deliver the read-backs the last cycle() left staged (logging only; no effect on any device value)"""
        for f in getattr(self, "_late", ()):
            f()
        self._late = []
        from . import ops
        ops.check_fused_sync()
