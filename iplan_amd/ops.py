"""Host-side launch wrappers: torch tensors in, C-ABI structs out.

Each function only (a) allocates outputs with torch, (b) fills the argument struct with raw device
pointers / strides, (c) calls the entry point on torch's current stream.  No arithmetic happens
here.  ``lib`` defaults to the gfx950 build; the CPU test-suite passes the host-emulated build of
the same kernel sources instead.
"""
import contextlib
import ctypes as C

import os

import torch

from . import _lib as L


def _lib(lib):
    return lib if lib is not None else L.get_lib()


def _nb_strides(t, inner):
    """t is logically [n_nets, B, <inner dims contiguous>]; returns (s_net, s_b) in elements."""
    assert t.dtype == torch.float32, t.dtype
    expect = 1
    for d in range(t.dim() - 1, 1, -1):
        assert t.stride(d) == expect or t.shape[d] == 1, (t.shape, t.stride())
        expect *= t.shape[d]
    assert expect == inner, (expect, inner)
    return t.stride(0), t.stride(1)


def _gat_flops(n_nets, B, N, D, A, H=32):
    """algorithmic FLOPs of one GAT forward: V (2DH + 24H^2 + 6HA + 12A^2) + P (12H^2 + 8H + 4A) per net"""
    V, P = B * N, B * N * (N - 1)
    return float(n_nets * (V * (2 * D * H + 24 * H * H + 6 * H * A + 12 * A * A) + P * (12 * H * H + 8 * H + 4 * A)))


def gat_forward(arena, src0, src1, h_prev, noise, tau=0.01, save=False, out=None, phase_clocks=None, lib=None, fuse_enc=None, fuse_ac=None):
    """GAT_Net.forward for all nets.  src0 [n_nets,B,N,d0], src1 [n_nets,B,N,d1] or None,
    h_prev [n_nets,B,N,A] (first two dims may be arbitrarily strided views), noise
    [n_nets,B,N,N-1,2] contiguous.  Returns (out [n_nets,B,N,A], saved dict or None).
    ``fuse_enc``: what ``enc_forward(..., launch=False)`` returned -- the encoder's latent update of the same rollout step rides
    in this launch (iplan_gat_enc_fwd: the GAT scenes' workgroups first, the encoder's behind them).
    ``fuse_ac`` (with ``fuse_enc``): what ``ac_forward(..., launch=False)`` returned for the NEXT step's action selection -- it
    reads what this launch writes and runs as its last workgroups (iplan_gat_enc_ac_fwd)."""
    lib = _lib(lib)
    n_nets, B, N, d0 = src0.shape
    d1 = 0 if src1 is None else src1.shape[-1]
    A = h_prev.shape[-1]
    dev = src0.device
    a = L.GatFwdArgs()
    a.n_nets, a.B, a.N, a.d0, a.d1 = n_nets, B, N, d0, d1
    a.src0 = src0.data_ptr()
    a.src0_s_net, a.src0_s_b = _nb_strides(src0, N * d0)
    if src1 is not None:
        a.src1 = src1.data_ptr()
        a.src1_s_net, a.src1_s_b = _nb_strides(src1, N * d1)
    a.h_prev = h_prev.data_ptr()
    a.h_s_net, a.h_s_b = _nb_strides(h_prev, N * A)
    if out is None:
        out = torch.empty(n_nets, B, N, A, dtype=torch.float32, device=dev)
    a.out = out.data_ptr()
    a.out_s_net, a.out_s_b = _nb_strides(out, N * A)
    assert noise.is_contiguous() and noise.shape == (n_nets, B, N, N - 1, 2), noise.shape
    a.noise = noise.data_ptr()
    a.params = arena.data.data_ptr()
    a.params_s_net = arena.net_stride
    for i, k in enumerate(L.GAT_PARAM_ORDER):
        a.off[i] = arena.off(k)
    a.tau = tau
    if phase_clocks is not None:
        assert phase_clocks.dtype == torch.int64 and phase_clocks.numel() >= n_nets * B * 5
        a.phase_clocks = phase_clocks.data_ptr()
    saved = None
    if save:
        H = A
        saved = dict(
            h_enc=torch.empty(n_nets, B * N, H, device=dev),
            gru=torch.empty(n_nets, 2, B, (N + 15) // 16, N - 1, L.GAT_REC_GROUPS, 16, 16, device=dev),   # tile-major 1 KiB blocks: h r z n (csrc/gat.hip, gat_bwd.hip: REC = 8 * 256 floats per (tile, step))
            qkv=torch.empty(n_nets, B * N, 3 * A, device=dev),
            soft=torch.empty(n_nets, B * N, N - 1, device=dev),
            hard=torch.empty(n_nets, B * N, N - 1, device=dev),
            x=torch.empty(n_nets, B * N, A, device=dev),
            cell=torch.empty(n_nets, B * N, 4 * A, device=dev),
        )
        for k, v in saved.items():
            setattr(a.saved, k, v.data_ptr())
    if fuse_enc is not None and fuse_ac is not None:
        sync = _fused_sync(dev)
        if TIMERS is not None and phase_clocks is None and TIMERS.sample("gat_scenes_clocks", 32):
            # bench.py's roofline: in-kernel stamps of the scenes (entry / end of every scene workgroup, wall_clock64 = 100 MHz) of
            # every 32nd fused launch -- the launch's own duration also holds the next step's action selection behind them
            clocks = torch.zeros(n_nets * B * 5, dtype=torch.int64, device=dev)
            a.phase_clocks = clocks.data_ptr()
            TIMERS.clocks.setdefault("gat_scenes_clocks", []).append(clocks)

        def fused3():
            rc = lib.c.iplan_gat_enc_ac_fwd(C.byref(a), C.byref(fuse_enc["args"]), C.byref(fuse_ac["_args"]), C.c_void_p(sync.data_ptr()),
                                            C.c_void_p(L.current_stream(dev) or 0))
            if rc != 0:
                raise L.IplanError(f"iplan_gat_enc_ac_fwd failed ({rc}): {lib.c.iplan_last_error().decode()}")
        _launch("gat_enc_ac_fwd_kernel", fused3)
    elif fuse_enc is not None:
        assert fuse_ac is None
        def fused():
            rc = lib.c.iplan_gat_enc_fwd(C.byref(a), C.byref(fuse_enc["args"]), C.c_void_p(L.current_stream(dev) or 0))
            if rc != 0:
                raise L.IplanError(f"iplan_gat_enc_fwd failed ({rc}): {lib.c.iplan_last_error().decode()}")
        _launch("gat_fwd_kernel", fused, work=_gat_flops(n_nets, B, N, d0 + d1, A))
    else:
        _launch("gat_fwd_kernel", lambda: lib.call("iplan_gat_fwd", a, L.current_stream(dev)), work=_gat_flops(n_nets, B, N, d0 + d1, A))
    if saved is not None:
        saved["_args"] = a
        saved["_keep"] = (src0, src1, h_prev, noise, out)
    return out, saved


_FUSED_SYNC = {}


def _fused_sync(dev):
    """the 3 counters of iplan_gat_enc_ac_fwd, private to (device, current stream)"""
    key = (str(dev), L.current_stream(dev))
    buf = _FUSED_SYNC.get(key)
    if buf is None:
        buf = _FUSED_SYNC[key] = torch.zeros(4, dtype=torch.int32, device=dev)
    return buf


def fused_sync_error():
    """True if any fused rollout launch gave up waiting for its producers (reads the device: tests / end of a rollout only)"""
    return any(int(b[2]) != 0 for b in _FUSED_SYNC.values())


def check_fused_sync():
    """Raise if a fused vector-step launch ran its action selection on stale latents (its wait hit the poll limit -- a scene or
    encoder workgroup never reported).  One 4-byte read-back per stream that ever ran a fused launch: call it where the host
    synchronises anyway (end of an episode / a training cycle)."""
    if fused_sync_error():
        for b in _FUSED_SYNC.values():                       # (a producer that reports after the give-up would leave the counters off by
            b.zero_()                                        #  one for every later launch: start the next one from a clean state)
        raise L.IplanError("iplan_gat_enc_ac_fwd: an action selection gave up waiting for the latent updates of its launch "
                           "(sync[2] != 0); the episode's actions from that step on are not valid")


def enc_forward(arena, x, h0, prev_latent, coef, Z, out_lat=None, out_h=None, lib=None, launch=True):
    """EncoderRNN + soft update for all nets.  x [n_nets,B,N,L,d] (any strides as long as the last dim is
    contiguous: a sliding window over a time-major observation log is read in place), h0 [n_nets,B,N,R],
    prev_latent [n_nets,B,N,Z] or None (first two dims may be strided views).  ``out_lat`` / ``out_h``:
    optional destinations of the same logical shapes (e.g. views into the episode buffer).
    Returns (latent [n_nets,B,N,Z], hL [n_nets,B,N,R])."""
    lib = _lib(lib)
    n_nets, B, N, Lw, d = x.shape
    R = h0.shape[-1]
    dev = x.device
    a = L.EncFwdArgs()
    a.n_nets, a.B, a.N, a.L, a.d, a.Z = n_nets, B, N, Lw, d, Z
    assert x.dtype == torch.float32 and x.stride(4) == 1
    a.x = x.data_ptr()
    a.x_s_net, a.x_s_b, a.x_s_i, a.x_s_t = x.stride(0), x.stride(1), x.stride(2), x.stride(3)
    a.h0 = h0.data_ptr()
    a.h0_s_net, a.h0_s_b = _nb_strides(h0, N * R)
    hL = out_h if out_h is not None else torch.empty(n_nets, B, N, R, dtype=torch.float32, device=dev)
    a.hL = hL.data_ptr()
    a.hL_s_net, a.hL_s_b = _nb_strides(hL, N * R)
    if prev_latent is not None:
        a.prev_latent = prev_latent.data_ptr()
        a.pl_s_net, a.pl_s_b = _nb_strides(prev_latent, N * Z)
    lat = out_lat if out_lat is not None else torch.empty(n_nets, B, N, Z, dtype=torch.float32, device=dev)
    a.latent_out = lat.data_ptr()
    a.lo_s_net, a.lo_s_b = _nb_strides(lat, N * Z)
    a.one_minus_c = 1.0 - coef
    a.c = coef
    a.params = arena.data.data_ptr()
    a.params_s_net = arena.net_stride
    for i, k in enumerate(L.ENC_PARAM_ORDER):
        a.off[i] = arena.off(k)
    if not launch:                                           # the caller hands ``args`` to gat_forward(fuse_enc=...)
        return lat, hL, dict(args=a, keep=(x, h0, prev_latent, lat, hL))
    lib.call("iplan_enc_fwd", a, L.current_stream(dev))
    return lat, hL


class AcFeatureSpec:
    """Where the actor/critic input row comes from (DcntrlMAC._build_inputs[_ippo] fused into the
    kernel).  ``sources``: up to three (tensor, width, s_net, s_row) per-entity fields; tensors are
    only referenced (kept alive) -- the kernel reads them in place."""

    def __init__(self, N, sources, n_actions=0, last_action=None, la_strides=(0, 0), n_id=0, T=1, T_phys=1):
        """last_action: int32 (or int64, read in place) tensor of hot indices (-1 = all zeros), or None."""
        self.N, self.sources, self.n_actions = N, sources, n_actions
        self.last_action, self.la_strides, self.n_id, self.T, self.T_phys = last_action, la_strides, n_id, T, T_phys

    @property
    def F(self):
        return self.N * sum(s[1] for s in self.sources) + self.n_actions + self.n_id

    def fill(self, f):
        f.N = self.N
        for k in range(3):
            if k < len(self.sources):
                t, w, s_net, s_row = self.sources[k]
                assert t.dtype == torch.float32
                f.w[k], f.src[k], f.s_net[k], f.s_row[k] = w, t.data_ptr(), s_net, s_row
            else:
                f.w[k], f.src[k], f.s_net[k], f.s_row[k] = 0, None, 0, 0
        f.n_actions = self.n_actions
        if self.last_action is not None:
            if self.last_action.dtype == torch.int64:
                f.last_action64 = self.last_action.data_ptr()
                f.la64_s_net, f.la64_s_row = self.la_strides
            else:
                assert self.last_action.dtype == torch.int32
                f.last_action = self.last_action.data_ptr()
                f.la_s_net, f.la_s_row = self.la_strides
        f.n_id = self.n_id
        f.T, f.T_phys = self.T, self.T_phys


def _fill_acnet(dst, arena, order, n_out):
    dst.params = arena.data.data_ptr()
    dst.params_s_net = arena.net_stride
    for i, k in enumerate(order):
        dst.off[i] = arena.off(k)
    dst.n_out = n_out


def ac_forward(actor_arena, critic_arena, which, spec, rows, n_agents, h_actor=None, h_critic=None,
               h_strides=(0, 0), avail=None, avail_strides=(0, 0), mode=0, q_noise=None, actions_in=None,
               act_strides=(0, 0), n_actions=5, ksplit=None, save=False, want_probs=False, want_entropy=False,
               want_h=True, h_out=None, actions_out=None, onehot_out=None, ln_stats=None, ln_stats_mode=0, phase_clocks=None,
               packed=None, xhat=None, lib=None, launch=True):
    """Fused actor (which=0) / critic (1) / both (2) forward for all agents.
    ``launch=False`` (rollout shape only): everything is set up but nothing is enqueued -- the returned dict goes to
    ``gat_forward(fuse_ac=...)``, whose launch runs this action selection behind the latent updates it depends on.
    Returns a dict with the requested outputs, each laid out [n_agents, rows, ...].
    Optional in-place destinations (rollout: write straight into the episode buffer):
      h_out = (actor_tensor, critic_tensor, (s_net, s_row));  actions_out = (int64 tensor, (s_net, s_row));
      onehot_out = (float tensor, (s_net, s_row)).
    ln_stats [n_agents, n_physical_rows, 2] with ln_stats_mode 1 (compute + store) / 2 (re-use).
    xhat: ``ac_xhat_pack(...)`` of the same (spec, rows) -- which = 2 only: fc1 runs as the split-bf16 contraction over the
    packed rows (iplan_ac_fc1_split_fwd) and this launch is the 64-wide tail; ``ac_backward`` then takes the matching path."""
    lib = _lib(lib)
    dev = (actor_arena if which != 1 else critic_arena).data.device
    a = L.AcFwdArgs()
    a.n_agents, a.rows, a.which = n_agents, rows, which
    a.ksplit = ksplit if ksplit is not None else (8 if rows <= 512 else 1)
    a.act_tanh = int(bool(getattr(actor_arena if which != 1 else critic_arena, "act_tanh", False)))
    if which == 2:                                            # one flag for both nets of the launch (and of its backward tail)
        assert bool(getattr(actor_arena, "act_tanh", False)) == bool(getattr(critic_arena, "act_tanh", False)), \
            "actor and critic of one fused launch must use the same trunk activation (args.use_ReLU)"
    spec.fill(a.feat)
    out = {}
    f32 = dict(dtype=torch.float32, device=dev)
    if which != 1:
        _fill_acnet(a.actor, actor_arena, L.ACTOR_PARAM_ORDER, n_actions)
        a.h_actor = h_actor.data_ptr()
        if h_out is not None:
            a.h_actor_out = h_out[0].data_ptr()
        elif want_h:
            out["h_actor"] = torch.empty(n_agents, rows, L.AC_HIDDEN, **f32)
            a.h_actor_out = out["h_actor"].data_ptr()
        if avail is not None:
            assert avail.dtype == torch.int32
            a.avail = avail.data_ptr()
            a.av_s_net, a.av_s_row = avail_strides
        a.mode = mode
        if mode == 1:
            assert q_noise.shape == (n_agents, rows, n_actions) and q_noise.is_contiguous()
            a.q_noise = q_noise.data_ptr()
        if mode == 2:
            assert actions_in.dtype == torch.int64
            a.actions_in = actions_in.data_ptr()
            a.act_s_net, a.act_s_row = act_strides
        elif actions_out is not None:
            assert actions_out[0].dtype == torch.int64
            a.actions_out = actions_out[0].data_ptr()
            a.ao_s_net, a.ao_s_row = actions_out[1]
        else:
            out["actions"] = torch.empty(n_agents, rows, dtype=torch.int64, device=dev)
            a.actions_out = out["actions"].data_ptr()
        if onehot_out is not None and mode != 2:
            assert onehot_out[0].dtype == torch.float32
            a.onehot_out = onehot_out[0].data_ptr()
            a.oh_s_net, a.oh_s_row = onehot_out[1]
        out["logp"] = torch.empty(n_agents, rows, **f32)
        a.logp = out["logp"].data_ptr()
        if want_entropy:
            out["entropy"] = torch.empty(n_agents, rows, **f32)
            a.entropy = out["entropy"].data_ptr()
        if want_probs:
            out["probs"] = torch.empty(n_agents, rows, n_actions, **f32)
            a.probs = out["probs"].data_ptr()
    if which != 0:
        _fill_acnet(a.critic, critic_arena, L.CRITIC_PARAM_ORDER, 1)
        a.h_critic = h_critic.data_ptr()
        if h_out is not None:
            a.h_critic_out = h_out[1].data_ptr()
        elif want_h:
            out["h_critic"] = torch.empty(n_agents, rows, L.AC_HIDDEN, **f32)
            a.h_critic_out = out["h_critic"].data_ptr()
        out["values"] = torch.empty(n_agents, rows, **f32)
        a.values = out["values"].data_ptr()
    a.hs_net, a.hs_row = h_strides
    if h_out is not None:
        a.ho_s_net, a.ho_s_row = h_out[2]
    if ln_stats is not None and ln_stats_mode:
        assert (ln_stats.dtype == torch.float32 and ln_stats.shape[0] == n_agents and ln_stats.shape[2] == 2
                and ln_stats.stride(2) == 1 and ln_stats.stride(1) == 2)          # rows contiguous (a row range of a larger buffer is fine)
        a.ln_stats, a.ln_stats_s_net, a.ln_stats_mode = ln_stats.data_ptr(), ln_stats.stride(0), ln_stats_mode
    if phase_clocks is not None:
        a.phase_clocks = phase_clocks.data_ptr()
    if packed is not None:                                 # Fc1Pack.get(): fragment-major fc1 operands of both arenas
        pa, pc = packed
        if a.ksplit > 1 and a.ln_stats_mode == 0:           # the folded form reads W gamma / W beta from the pack
            assert getattr(packed, "fold", False), "rollout-shaped launch needs Fc1Pack.get(spec, fold=True)"
        if which != 1:
            a.packed_actor, a.packed_s_net = pa.data_ptr(), pa.stride(0)
        if which != 0:
            a.packed_critic, a.packed_s_net = pc.data_ptr(), pc.stride(0)
    if save:
        out["saved"] = torch.empty(2, n_agents, rows, L.AC_SAVE_FLOATS, **f32)
        a.saved = out["saved"].data_ptr()
    ks_bufs = None
    units = ((rows + 15) // 16) * n_agents * (2 if which == 2 else 1)
    # rollout shape: the F-wide contraction of every (row tile, net) unit over kw workgroups (include/iplan_hip.h: ksplit_wg)
    # while the launch is far from filling the 256 CUs: 4 up to 40 units (config 3: 20), 2 up to 96, off beyond
    kw = int(os.environ.get("IPLAN_AC_KSPLIT_WG", "0")) or (4 if units <= 40 else (2 if units <= 96 else 1))
    if (kw > 1 and a.ksplit == 8 and not save and a.ln_stats_mode == 0 and packed is not None and getattr(packed, "fold", False)):
        key = (str(dev), L.current_stream(dev), units, kw)
        ks_bufs = _KSPLIT_BUFS.get(key)
        if ks_bufs is None:
            ks_bufs = _KSPLIT_BUFS[key] = (torch.empty(units * kw * L.AC_KS_SLOT_FLOATS, **f32),
                                           torch.zeros(units, dtype=torch.int32, device=dev))
        a.ksplit_wg, a.ks_scratch, a.ks_count = kw, ks_bufs[0].data_ptr(), ks_bufs[1].data_ptr()
    z1 = None
    if xhat is not None:
        assert which == 2 and a.ksplit == 1 and ln_stats_mode == 2 and xhat["rows"] == rows and xhat["n_agents"] == n_agents
        s = L.AcFc1SplitArgs()
        s.n_agents, s.rows = n_agents, rows
        spec.fill(s.feat)
        s.actor, s.critic = a.actor, a.critic
        KS = (xhat["KT"] + 1) // 2
        ws = workspace(dev, n_agents * KS * 6144 + 2 * n_agents * L.AC_HIDDEN, "fc1_split")
        kparts = int(lib.c.iplan_ac_fc1_split_parts(n_agents, rows))           # > 1: small batches, the K loop over several workgroups
        z1 = torch.empty(kparts, 2, n_agents, rows, L.AC_HIDDEN, **f32)
        s.kparts, a.fc1_pre_parts = kparts, kparts
        s.xf, s.wsplit, s.wbeta, s.z1 = xhat["xf"].data_ptr(), ws.data_ptr(), ws.data_ptr() + 4 * n_agents * KS * 6144, z1.data_ptr()
        # algorithmic work of the launch: 2 FLOP per (row, feature, output) of both nets
        _launch("ac_fc1_split_fwd", lambda: lib.call("iplan_ac_fc1_split_fwd", s, L.current_stream(dev)),
                work=2.0 * n_agents * rows * spec.F * 2 * L.AC_HIDDEN)
        a.fc1_pre = z1.data_ptr()
        out["_xhat"] = xhat
    if launch:
        _launch("ac_fwd_kernel:train" if save else ("ac_fwd_kernel:rollout" if rows <= 512 else "ac_fwd_kernel:infer"),
                lambda: lib.call("iplan_ac_fwd", a, L.current_stream(dev)))
    else:
        assert a.ksplit == 8 and not save and xhat is None and a.ln_stats_mode == 0, "launch=False is for the rollout-shaped forward"
    out["_args"] = a
    out["_keep"] = (spec, h_actor, h_critic, avail, q_noise, actions_in, h_out, actions_out, onehot_out, ln_stats, packed, z1, xhat, ks_bufs)
    return out


def ac_xhat_pack(spec, rows, n_agents, ln_stats, lib=None):
    """The normalised feature rows of a PPO batch, gathered once per train() into the two fragment-major copies the
    split-bf16 fc1 kernels stream (include/iplan_hip.h: IplanAcXhatArgs).  ``ln_stats``: the (mean, rstd) table a
    ``ac_forward(..., ln_stats_mode=1)`` pass stored for the same physical rows."""
    lib = _lib(lib)
    dev = ln_stats.device
    a = L.AcXhatArgs()
    a.n_agents, a.rows = n_agents, rows
    spec.fill(a.feat)
    assert ln_stats.dtype == torch.float32 and ln_stats.shape[0] == n_agents and ln_stats.stride(2) == 1 and ln_stats.stride(1) == 2
    a.ln_stats, a.ln_stats_s_net = ln_stats.data_ptr(), ln_stats.stride(0)
    nf, nb = (int(lib.c.iplan_ac_xhat_floats(C.byref(a.feat), rows, k)) for k in (0, 1))
    xf = torch.empty(n_agents, nf, dtype=torch.float32, device=dev)
    xb = torch.empty(n_agents, nb, dtype=torch.float32, device=dev)
    a.xf, a.xb = xf.data_ptr(), xb.data_ptr()
    _launch("ac_xhat_pack", lambda: lib.call("iplan_ac_xhat_pack", a, L.current_stream(dev)))
    return dict(xf=xf, xb=xb, rows=rows, n_agents=n_agents, KT=int(lib.c.iplan_ac_kpad(C.byref(a.feat))) // 16, _keep=(spec, ln_stats))


class _Packed(tuple):
    """(actor buffer, critic buffer) of Fc1Pack.get; ``fold`` says whether the W gamma / W beta tail is current"""
    fold = False


class Fc1Pack:
    """fc1.weight + feature_norm.{weight, bias} of the actor and critic arenas in the forward kernels' own K order and MFMA
    fragment order (iplan_ac_pack_fc1).  ``get(spec)`` repacks only when an arena changed since the last pack or the feature
    layout differs.  What counts as "changed" (the staleness key):
      * ``arena.version`` -- bumped by every kernel-side write (FusedAdam, DataParallel.broadcast_arena), by the modules'
        ``load_state_dict`` hook and by ``arena.touch()``;
      * the sum of ``p._version`` over the packed Parameters of all nets -- in-place writes THROUGH a Parameter (``p.add_()``,
        ``p.copy_()``, ``nn.init.*`` under no_grad, a torch.optim optimiser on ``mac.parameters()``) bump it.  The Parameters are
        bound to the arena with ``p.data = view``, which does NOT share the arena tensor's version counter, so
        ``arena.data._version`` says nothing about them;
      * ``arena.data._version`` -- writes on the arena tensor itself.
    A write through a detached alias (``p.data.mul_()``, a raw pointer) is invisible to all three: call ``arena.touch()``.
    ``fold=True`` also brings W gamma / W beta up to date (the rollout's folded LayerNorm(F) reads them; the streaming
    forward of a PPO epoch does not, and a repack after each of its 30 Adam steps would spend more on them than on the
    fragments)."""

    def __init__(self, actor_arena, critic_arena):
        self.arenas = (actor_arena, critic_arena)
        names = ("base.mlp.fc1.0.weight", "base.feature_norm.weight", "base.feature_norm.bias")
        self.watched = [[dict(m.named_parameters())[n] for m in arena.modules for n in names] for arena in self.arenas]
        self.buf = [None, None]
        self.key = [[None, None], [None, None]]             # per arena: key of the fragment part, of the W gamma / W beta part

    def get(self, spec, fold=False, lib=None):
        if os.environ.get("IPLAN_NO_FC1_PACK"):             # A/B knob: read the arena in place (scattered fragment loads)
            return None
        lib = _lib(lib)
        a = L.AcPackArgs()
        spec.fill(a.feat)
        floats = int(lib.c.iplan_ac_packed_floats(C.byref(a.feat)))
        sig = (spec.N, tuple(s[1] for s in spec.sources), spec.n_actions, spec.n_id)
        for k, (arena, order) in enumerate(zip(self.arenas, (L.ACTOR_PARAM_ORDER, L.CRITIC_PARAM_ORDER))):
            key = (arena.data._version, arena.version, sum(p._version for p in self.watched[k]), sig)
            if self.buf[k] is None or self.buf[k].shape[1] != floats:
                self.buf[k] = torch.empty(arena.n_nets, floats, dtype=torch.float32, device=arena.data.device)
                self.key[k] = [None, None]
            todo = [p for p, need in ((0, True), (1, fold)) if need and self.key[k][p] != key]
            if not todo:
                continue
            a.n_nets = arena.n_nets
            a.params, a.params_s_net = arena.data.data_ptr(), arena.net_stride
            a.off_w1, a.off_fn_w, a.off_fn_b = (arena.off(n) for n in ("base.mlp.fc1.0.weight", "base.feature_norm.weight", "base.feature_norm.bias"))
            a.packed, a.packed_s_net = self.buf[k].data_ptr(), floats
            a.parts = 0 if len(todo) == 2 else todo[0] + 1
            lib.call("iplan_ac_pack_fc1", a, L.current_stream(arena.data.device))
            for p in todo:
                self.key[k][p] = key
        out = _Packed((self.buf[0], self.buf[1]))
        out.fold = all(self.key[k][1] == self.key[k][0] for k in range(2))
        return out


# ---- weight gradients ----------------------------------------------------------------------------------
_WORKSPACES = {}
_SIDE_STREAMS = {}

def _side_stream(dev, *key):
    """one side stream per (device, caller's stream, tag), created on first use (dict.setdefault would construct -- and
    discard -- a stream per call and walk torch's 32-entry stream pool into aliasing streams already held)"""
    k = (str(dev),) + key
    s = _SIDE_STREAMS.get(k)
    if s is None:
        # side streams exist to run BESIDE the caller's stream: not on its hardware queue, and the encoder's BPTT stream (tag 2) not on
        # the forward / in-line stream's either (streams.distinct_stream: probed once, at creation)
        from .streams import distinct_stream, probe_mode
        avoid = [torch.cuda.current_stream(dev)]
        if probe_mode() == "full" and len(key) > 1 and key[1] == 2:
            avoid.append(_SIDE_STREAMS.get((str(dev), key[0])))
        s = _SIDE_STREAMS[k] = distinct_stream(dev, avoid)
    return s

_QUEUES_VERIFIED = set()


def verify_side_queues(dev, main, extra=None):
    """IPLAN_QUEUE_PROBE=verify (streams.probe_mode), once per (device, main stream), AFTER the side streams had their first use: the
    encoder's forward / BPTT side streams of ``main`` -- and the caller's ``extra`` roles {name: stream} (harness: the prediction
    learner's) -- must not share ``main``'s hardware queue nor each other's (the encoder's two may: they never run at the same time);
    one that does is replaced.  -> (list of replaced role names, {name: stream now in use})."""
    from .streams import distinct_stream, probe_mode, shares_queue
    key = (str(dev), main.cuda_stream)
    roles = dict(extra or {})
    k_f, k_b = (str(dev), main.cuda_stream), (str(dev), main.cuda_stream, 2)
    roles.update(enc_fwd=_SIDE_STREAMS.get(k_f), enc_bwd=_SIDE_STREAMS.get(k_b))
    if key in _QUEUES_VERIFIED and not extra:
        return [], roles
    _QUEUES_VERIFIED.add(key)
    if probe_mode() != "verify" or torch.device(dev).type != "cuda" or not hasattr(torch.cuda, "_sleep"):
        return [], roles
    replaced = []
    apart = lambda a, b: {a, b} != {"enc_fwd", "enc_bwd"}    # noqa: E731
    try:
        torch.cuda.synchronize(dev)
        for name in ["enc_bwd", "enc_fwd"] + [n for n in roles if n not in ("enc_bwd", "enc_fwd")]:
            s = roles.get(name)
            if s is None:
                continue
            keep_off = [main] + [o for n, o in roles.items() if n != name and o is not None and apart(name, n)]
            if any(shares_queue(o, s, dev) for o in keep_off):
                roles[name] = distinct_stream(dev, keep_off, force=True)
                replaced.append(name)
                if name == "enc_fwd":
                    _SIDE_STREAMS[k_f] = roles[name]
                elif name == "enc_bwd":
                    _SIDE_STREAMS[k_b] = roles[name]
        torch.cuda.synchronize(dev)
    except Exception as e:                                   # noqa: BLE001 -- never let the check cost the run
        import sys
        print(f"[iplan_amd] hardware-queue check skipped ({type(e).__name__}: {str(e)[:100]})", file=sys.stderr)
    return replaced, roles


_KSPLIT_BUFS = {}


class KernelTimers:
    """In-situ kernel timing for bench.py's roofline: HIP events recorded on the launch stream right around selected
    launches (every ``every[name]``-th one) while the real workload runs.  ``ops.TIMERS = KernelTimers(...)`` turns
    it on; ``summary()`` (after a device synchronise) gives {name: (launches timed, mean seconds, mean work)} where
    ``work`` is whatever the call site passed (algorithmic bytes / FLOPs of that launch; 0 if it passed nothing)."""

    def __init__(self, every=None):
        self.every = dict(every or {})
        self.count = {}
        self.spans = {}
        self.clocks = {}                                    # name -> device buffers of in-kernel stamps (read after a synchronise)

    def sample(self, name, every):
        n = self.count.get(name, 0)
        self.count[name] = n + 1
        return n % every == 0

    def launch(self, name, fn, stream=None, work=0.0):
        n = self.count.get(name, 0)
        self.count[name] = n + 1
        if n % self.every.get(name, 1):
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        fn()
        e1.record(stream)
        self.spans.setdefault(name, []).append((e0, e1, float(work)))

    def summary(self):
        return {k: (len(v), sum(a.elapsed_time(b) for a, b, _ in v) / len(v) / 1e3, sum(w for _, _, w in v) / len(v))
                for k, v in self.spans.items()}


TIMERS = None


def _launch(name, fn, stream=None, work=0.0):
    """``stream``: the torch stream the launch goes to when it is not the current one (events must sit on that stream)."""
    if TIMERS is None:
        return fn()
    return TIMERS.launch(name, fn, stream, work)


def workspace(device, floats, tag="wgrad"):
    """Grow-only scratch buffer per (device, current stream, tag) -- owned by torch's caching allocator.  Keyed by
    stream because independent learners may run concurrently on different HIP streams."""
    sid = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    key = (str(device), sid, tag)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(max(int(floats), 1), dtype=torch.float32, device=device)
        _WORKSPACES[key] = buf
    return buf


class Wgrad:
    """Batch of dW = dY^T X problems (include/iplan_hip.h: IplanWgradProblem) written into a gradient arena."""

    def __init__(self, grad, n_nets, tag="iplan_wgrad"):
        self.grad, self.n_nets = grad, n_nets
        self.problems = []
        self._keep = []
        self.tag = tag                                      # KernelTimers key of the launches (bench.py's roofline entries)

    def add(self, dy, dy_strides, O, n_outer, n_inner, x=None, x_strides=(0, 0, 0), K=0, dw_off=-1, db_off=-1,
            dw_ld=None, dw_col0=0, seg=None, x_col0=0, x_shift=0, x0=None, x0_strides=(0, 0), beta=0.0, scale=1.0,
            dy_cg_stride=0, x_cg_stride=0, x_pre_valid=False):
        p = L.WgradProblem()
        p.dy = dy.data_ptr() if torch.is_tensor(dy) else dy
        p.dy_s_net, p.dy_s_outer, p.dy_s_inner = dy_strides
        if x is not None:
            p.x = x.data_ptr() if torch.is_tensor(x) else x
            p.x_s_net, p.x_s_outer, p.x_s_inner = x_strides
        if x0 is not None:
            p.x0 = x0.data_ptr() if torch.is_tensor(x0) else x0
            p.x0_s_net, p.x0_s_outer = x0_strides
        p.dw_off, p.db_off = dw_off, db_off
        p.O, p.K = O, K
        p.seg_split, p.seg_c0, p.seg_c1 = seg if seg is not None else (O, 0, 0)
        p.x_col0, p.x_shift, p.n_outer, p.n_inner = x_col0, x_shift, n_outer, n_inner
        p.dw_ld = K if dw_ld is None else dw_ld
        p.dw_col0 = dw_col0
        p.beta, p.scale = beta, scale
        p.dy_cg_stride, p.x_cg_stride, p.x_pre_valid = dy_cg_stride, x_cg_stride, int(bool(x_pre_valid))
        self.problems.append(p)
        self._keep += [dy, x, x0]
        return self

    def run(self, lib=None):
        lib = _lib(lib)
        dev = self.grad.device
        stream = L.current_stream(dev)
        for i0 in range(0, len(self.problems), L.WGRAD_MAX):
            chunk = self.problems[i0:i0 + L.WGRAD_MAX]
            a = L.WgradArgs()
            a.n_problems, a.n_nets = len(chunk), self.n_nets
            a.grad = self.grad.data_ptr()
            a.grad_s_net = self.grad.stride(0)
            for i, p in enumerate(chunk):
                a.p[i] = p
            need = lib.c.iplan_wgrad_workspace_floats(C.byref(a))
            ws = workspace(dev, need)
            a.workspace, a.workspace_floats = ws.data_ptr(), ws.numel()
            # algorithmic HBM bytes: every operand row of every problem read exactly once (4 (O + K) bytes per row) -- with the dY columns
            # two problems SHARE counted once (the two weights of a GRU: [dr dz] of the same rows; wgrad.hip pairs those jobs and does
            # read them once since round 5, so the figure is the least any kernel could move, not the least THIS kernel moves)
            nbytes = 4.0 * self.n_nets * sum(p.n_outer * p.n_inner * (p.O + p.K) for p in chunk)
            seen = {}
            for p in chunk:
                if p.O == 192 and p.K == 64 and p.seg_split >= 128:
                    key = (p.dy, p.dy_s_outer, p.dy_s_inner, p.n_outer, p.n_inner, p.seg_c0)
                    if key in seen:
                        nbytes -= 4.0 * self.n_nets * p.n_outer * p.n_inner * 128
                        del seen[key]
                    else:
                        seen[key] = True
            _launch(self.tag, lambda: lib.call("iplan_wgrad", a, stream), work=nbytes)


def _ac_wgrad(w, arena, head_names, saved, dsave, which, n_agents, rows, T, h, h_strides, T_phys, n_out, tiles, ln_part):
    """Weight-gradient problems of the 64-wide layers of one arena (actor or critic)."""
    M, SF, DS = L.AC_HIDDEN, L.AC_SAVE_FLOATS, L.AC_DSAVE_FLOATS
    sv = saved[which]                     # [n_agents, rows, SF]
    ds = dsave[which]                     # [n_agents, rows, DS]
    dptr, sptr = ds.data_ptr(), sv.data_ptr()
    dst = (rows * DS, DS, DS)
    sst = (rows * SF, SF, SF)
    off = arena.off
    # head: dY = dhead, X = f3 (slot 9)
    w.add(dptr + 4 * 6 * M, dst, n_out, rows, 1, x=sptr + 4 * 9 * M, x_strides=sst, K=M,
          dw_off=off(head_names[0]), db_off=off(head_names[1]))
    # GRU input weights: dgi = [dr, dz, dn_i] (cols 0..3M of the gate block), X = f2 (slot 3).  Rows indexed (episode, step) like the
    # hidden weights' problem below -- same dY pointer, strides and row split -- so that iplan_wgrad runs the two as ONE paired launch that
    # fetches [dr dz] once (wgrad.hip: wgrad_pair_bf16_kernel)
    n_ep = rows // T
    assert n_ep * T == rows
    w.add(dptr + 4 * 2 * M, (rows * DS, T * DS, DS), 3 * M, n_ep, T, x=sptr + 4 * 3 * M, x_strides=(rows * SF, T * SF, SF), K=M,
          dw_off=off("rnn.rnn.weight_ih_l0"), db_off=off("rnn.rnn.bias_ih_l0"))
    # GRU hidden weights: dgh = [dr, dz, dn_h], X = the stored hidden state of the row (episode layout)
    w.add(dptr + 4 * 2 * M, (rows * DS, T * DS, DS), 3 * M, n_ep, T, x=h, x_strides=(h_strides[0], T_phys * h_strides[1], h_strides[1]),
          K=M, dw_off=off("rnn.rnn.weight_hh_l0"), db_off=off("rnn.rnn.bias_hh_l0"), seg=(2 * M, 0, 3 * M))
    # fc2: dY = dz2, X = f1 (slot 1)
    w.add(dptr + 4 * M, dst, M, rows, 1, x=sptr + 4 * M, x_strides=sst, K=M,
          dw_off=off("base.mlp.fc2.0.0.weight"), db_off=off("base.mlp.fc2.0.0.bias"))
    # fc1 bias (S of the fc1 finalize)
    w.add(dptr, dst, M, rows, 1, db_off=off("base.mlp.fc1.0.bias"))
    # LayerNorm affine parameters: column sums of the per-tile partials
    lp = ln_part[which]                   # [n_agents, tiles, 6M]
    lst = (tiles * 6 * M, 6 * M, 6 * M)
    for k, name in enumerate(("rnn.norm", "base.mlp.fc2.0.2", "base.mlp.fc1.2")):
        w.add(lp.data_ptr() + 4 * (2 * k) * M, lst, M, tiles, 1, db_off=off(name + ".weight"))
        w.add(lp.data_ptr() + 4 * (2 * k + 1) * M, lst, M, tiles, 1, db_off=off(name + ".bias"))
    w._keep += [sv, ds, lp]


def n_which_rows(which, n_agents, rows):
    """rows x agents x nets of an actor/critic launch (which = 2: both nets)"""
    return (2 if which == 2 else 1) * n_agents * rows


def ac_backward(fwd, actor_arena, critic_arena, g_logp=None, g_entropy=0.0, g_values=None, lib=None):
    """Backward of an ``ac_forward(..., mode=2, save=True)`` launch: fills the gradient arenas of the
    nets that took part.  g_logp / g_values [n_agents, rows]; g_entropy: per-row tensor or a constant."""
    lib = _lib(lib)
    fa = fwd["_args"]
    which, n_agents, rows = fa.which, fa.n_agents, fa.rows
    spec, h_actor, h_critic = fwd["_keep"][0], fwd["_keep"][1], fwd["_keep"][2]
    saved = fwd["saved"]
    dev = saved.device
    f32 = dict(dtype=torch.float32, device=dev)
    a = L.AcBwdArgs()
    a.fwd = fa
    if which != 1:
        assert g_logp.shape == (n_agents, rows) and g_logp.is_contiguous()
        a.g_logp = g_logp.data_ptr()
        if torch.is_tensor(g_entropy):
            assert g_entropy.shape == (n_agents, rows) and g_entropy.is_contiguous()
            a.g_entropy = g_entropy.data_ptr()
        else:
            a.g_entropy_const = float(g_entropy)
        a.actor_grad = actor_arena.grad.data_ptr()
        a.actor_grad_s_net = actor_arena.grad.stride(0)
    if which != 0:
        assert g_values.shape == (n_agents, rows) and g_values.is_contiguous()
        a.g_values = g_values.data_ptr()
        a.critic_grad = critic_arena.grad.data_ptr()
        a.critic_grad_s_net = critic_arena.grad.stride(0)
    tiles = (rows + 15) // 16
    dsave = torch.empty(2, n_agents, rows, L.AC_DSAVE_FLOATS, **f32)
    ln_part = torch.empty(2, n_agents, tiles, L.AC_LNPART_FLOATS, **f32)
    a.dsave, a.ln_part = dsave.data_ptr(), ln_part.data_ptr()
    Fpad = int(lib.c.iplan_ac_kpad(C.byref(fa.feat)))
    n_which = 2 if which == 2 else 1
    xhat = fwd.get("_xhat")
    if xhat is not None:
        # split-bf16 form: a workgroup owns a block of k-tiles x both nets over a row chunk; the library sizes the chunks
        cr = C.c_int32(0)
        lib.c.iplan_ac_fc1_split_chunks(C.byref(fa.feat), n_agents, rows, C.byref(cr))
        chunk_rows = cr.value
        a.xb = xhat["xb"].data_ptr()
    else:
        # the fc1 contraction runs two waves per SIMD (8 k-tiles x 4 o-tiles per wave): as many row chunks as fill the chip once
        k_groups = int(lib.c.iplan_ac_fc1_groups(C.byref(fa.feat)))
        want = max(1, 2048 // (k_groups * n_which * n_agents))
        chunk_rows = max(256, ((rows + want - 1) // want + 15) // 16 * 16)
    chunks = (rows + chunk_rows - 1) // chunk_rows
    g_part = workspace(dev, n_which * n_agents * chunks * L.AC_HIDDEN * Fpad, "fc1")
    a.g_part, a.fc1_chunk_rows, a.fc1_chunks = g_part.data_ptr(), chunk_rows, chunks
    stream = L.current_stream(dev)
    # (algorithmic bytes: the record slots the tail reads -- a1, a2, the four gate rows, h', f3, the statistics -- + the row's hidden state,
    # the row gradients and the per-tile LayerNorm partials written)
    _launch("ac_bwd_tail_kernel", lambda: lib.call("iplan_ac_bwd_tail", a, stream),
            work=4.0 * n_which_rows(which, n_agents, rows) * (8 * L.AC_HIDDEN + 8 + L.AC_HIDDEN + L.AC_DSAVE_FLOATS + L.AC_LNPART_FLOATS / 16.0))
    ev_tail = None
    if which == 2 and dev.type == "cuda":
        ev_tail = torch.cuda.Event()
        ev_tail.record(torch.cuda.current_stream(dev))
    if xhat is not None:
        _launch("ac_fc1_split_wgrad", lambda: lib.call("iplan_ac_bwd_fc1_split", a, stream), work=2.0 * n_agents * rows * spec.F * 2 * L.AC_HIDDEN)
    else:
        lib.call("iplan_ac_bwd_fc1", a, stream)
    T, T_phys = spec.T, spec.T_phys
    hs = (fa.hs_net, fa.hs_row)
    # The weight-gradient contractions of the 64-wide layers are small, latency-bound launches (a wide one and three narrow
    # ones per arena, 0.25 ms per arena at config 3) and the two arenas are independent: the critic's run on a side stream
    # beside the actor's and the fc1 contraction, joined before the finalize (it reads both fc1.bias gradients).
    side = None
    if which == 2 and dev.type == "cuda" and not os.environ.get("IPLAN_AC_WGRAD_SERIAL"):
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev, main.cuda_stream, "ac")
        side.wait_event(ev_tail)
    if which != 1:
        w = Wgrad(actor_arena.grad, n_agents)
        _ac_wgrad(w, actor_arena, ("act.action_out.linear.weight", "act.action_out.linear.bias"), saved, dsave, 0,
                  n_agents, rows, T, h_actor.data_ptr(), hs, T_phys, fa.actor.n_out, tiles, ln_part)
        w.run(lib)
    if which != 0:
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            w = Wgrad(critic_arena.grad, n_agents)
            _ac_wgrad(w, critic_arena, ("v_out.weight", "v_out.bias"), saved, dsave, 1,
                      n_agents, rows, T, h_critic.data_ptr(), hs, T_phys, 1, tiles, ln_part)
            w.run(lib)
            if side is not None:
                ev_side = torch.cuda.Event()
                ev_side.record(side)
        if side is not None:
            main.wait_event(ev_side)
    lib.call("iplan_ac_bwd_fc1_finalize", a, stream)
    return dict(dsave=dsave, ln_part=ln_part)


def gat_backward(arena, saved, g_out, phase_clocks=None, lib=None):
    """Backward of a ``gat_forward(..., save=True)`` launch.  g_out [n_nets, B, N, A] (first two dims may
    be strided).  Fills arena.grad (every GAT parameter of every net)."""
    lib = _lib(lib)
    fa = saved["_args"]
    src0, src1, h_prev = saved["_keep"][0], saved["_keep"][1], saved["_keep"][2]
    n_nets, B, N, d0, d1 = fa.n_nets, fa.B, fa.N, fa.d0, fa.d1
    H = A = 32
    dev = g_out.device
    f32 = dict(dtype=torch.float32, device=dev)
    a = L.GatBwdArgs()
    a.fwd = fa
    a.g_out = g_out.data_ptr()
    a.g_s_net, a.g_s_b = _nb_strides(g_out, N * A)
    dgru = torch.empty(n_nets, 2, B, (N + 15) // 16, N - 1, 2, 3 * H, **f32)           # kernel-private scratch (class rows per tile and step)
    whh_part = torch.empty(n_nets, B, 2, 4, L.GAT_WHH_PART, **f32)
    a.whh_part, a.grad, a.grad_s_net = whh_part.data_ptr(), arena.grad.data_ptr(), arena.grad.stride(0)
    node_dy = torch.empty(n_nets, B * N, L.GAT_NODE_DY, **f32)
    hard_part = torch.empty(n_nets, B, L.GAT_HARD_PART, **f32)
    a.dgru, a.node_dy, a.hard_part = dgru.data_ptr(), node_dy.data_ptr(), hard_part.data_ptr()
    if phase_clocks is not None:                            # int64 [>= 15]: workgroup 0's clocks land in slots 8..14
        assert phase_clocks.dtype == torch.int64 and phase_clocks.numel() >= 15
        a.fwd.phase_clocks = phase_clocks.data_ptr()
    _launch("gat_bwd_kernel", lambda: lib.call("iplan_gat_bwd", a, L.current_stream(dev)))
    a.fwd.phase_clocks = fa.phase_clocks

    w = Wgrad(arena.grad, n_nets)
    off = arena.off
    DYW = L.GAT_NODE_DY
    nd = node_dy.data_ptr()
    nst = (B * N * DYW, N * DYW, DYW)                       # rows (b, i)
    D = d0 + d1
    # encoding: dY = ENC, X = [src0 || src1] rows
    w.add(nd, nst, H, B, N, x=src0, x_strides=(src0.stride(0), src0.stride(1), d0), K=d0,
          dw_off=off("encoding.weight"), dw_ld=D, db_off=off("encoding.bias"))
    if d1 > 0:
        w.add(nd, nst, H, B, N, x=src1, x_strides=(src1.stride(0), src1.stride(1), d1), K=d1,
              dw_off=off("encoding.weight"), dw_ld=D, dw_col0=d0)
    he = saved["h_enc"]
    hst = (B * N * H, N * H, H)
    gru = saved["gru"]
    P1 = N - 1
    for dr, sfx in ((0, ""), (1, "_reverse")):
        # separable input projection: W_ih = [W_a | W_b]
        w.add(nd + 4 * (32 + dr * 192), nst, 3 * H, B, N, x=he, x_strides=hst, K=H,
              dw_off=off("hard_bi_GRU.weight_ih_l0" + sfx), dw_ld=2 * H, db_off=off("hard_bi_GRU.bias_ih_l0" + sfx))
        w.add(nd + 4 * (128 + dr * 192), nst, 3 * H, B, N, x=he, x_strides=hst, K=H,
              dw_off=off("hard_bi_GRU.weight_ih_l0" + sfx), dw_ld=2 * H, dw_col0=H)
        # (recurrent weights hard_bi_GRU.weight_hh / bias_hh: accumulated by the backward kernel itself)
        # hard_encoding.weight[c][dr*H : (dr+1)*H] = -/+ sum dDelta * h  (per-scene partials from the kernel)
        for c, sc in ((0, -1.0), (1, 1.0)):
            w.add(hard_part.data_ptr() + 4 * dr * 4 * H, (B * L.GAT_HARD_PART, L.GAT_HARD_PART, H), H, B, 4,
                  db_off=off("hard_encoding.weight") + c * 2 * H + dr * H, scale=sc)
    for c, sc in ((0, -1.0), (1, 1.0)):
        w.add(hard_part.data_ptr() + 4 * 8 * H, (B * L.GAT_HARD_PART, L.GAT_HARD_PART, L.GAT_HARD_PART), 1, B, 1,
              db_off=off("hard_encoding.bias") + c, scale=sc)
    w.add(nd + 4 * 416, nst, A, B, N, x=he, x_strides=hst, K=H, dw_off=off("q.weight"))
    w.add(nd + 4 * 448, nst, A, B, N, x=he, x_strides=hst, K=H, dw_off=off("k.weight"))
    w.add(nd + 4 * 480, nst, A, B, N, x=he, x_strides=hst, K=H, dw_off=off("v.weight"), db_off=off("v.bias"))
    xs = saved["x"]
    w.add(nd + 4 * 512, nst, 3 * A, B, N, x=xs, x_strides=(B * N * A, N * A, A), K=A,
          dw_off=off("rnn.weight_ih"), db_off=off("rnn.bias_ih"))
    w.add(nd + 4 * 512, nst, 3 * A, B, N, x=h_prev, x_strides=(h_prev.stride(0), h_prev.stride(1), A), K=A,
          dw_off=off("rnn.weight_hh"), db_off=off("rnn.bias_hh"), seg=(2 * A, 0, 3 * A))
    w._keep += [dgru, node_dy, hard_part, saved]
    w.run(lib)
    return dict(dgru=dgru, node_dy=node_dy, hard_part=hard_part)


# ---- prediction decoder --------------------------------------------------------------------------------
def pdec_forward(arena, x0, h0, target, mask, N, keep=None, drop_p=0.0, teacher=None, mask_sum=None, lib=None):
    """Prediction_Decoder.forward + masked-L1 loss for all nets.  x0 [n_nets, rows, d], h0 [n_nets, rows, 32],
    target [n_nets, rows, P, d], mask [n_nets, rows // N], keep [n_nets, P, rows, 32] or None, teacher int32
    [n_nets, P] or None.  Returns dict(pred, loss [n_nets], saved, ...)."""
    lib = _lib(lib)
    n_nets, rows, d = x0.shape
    P = target.shape[2]
    dev = x0.device
    f32 = dict(dtype=torch.float32, device=dev)
    for t in (x0, h0, target, mask):
        assert t.is_contiguous() and t.dtype == torch.float32
    assert h0.shape == (n_nets, rows, 32) and target.shape == (n_nets, rows, P, d) and mask.shape == (n_nets, rows // N)
    a = L.PdecArgs()
    a.n_nets, a.rows, a.N, a.P, a.d = n_nets, rows, N, P, d
    a.x0, a.h0, a.target, a.mask = x0.data_ptr(), h0.data_ptr(), target.data_ptr(), mask.data_ptr()
    if keep is not None:
        assert keep.shape == (n_nets, P, rows, 32) and keep.is_contiguous() and keep.dtype == torch.float32
        a.keep = keep.data_ptr()
    a.drop_p = drop_p
    if teacher is not None:
        assert teacher.dtype == torch.int32 and teacher.shape == (n_nets, P) and teacher.is_contiguous()
        a.teacher = teacher.data_ptr()
    a.params, a.params_s_net = arena.data.data_ptr(), arena.net_stride
    for i, k in enumerate(L.DEC_PARAM_ORDER):
        a.off[i] = arena.off(k)
    if mask_sum is not None:                               # loss normaliser over all data-parallel ranks' samples
        assert mask_sum.shape == (n_nets,) and mask_sum.dtype == torch.float32 and mask_sum.is_contiguous()
        a.mask_sum = mask_sum.data_ptr()
    tiles = (rows + 15) // 16
    out = dict(pred=torch.empty(n_nets, rows, P, d, **f32), saved=torch.empty(n_nets, rows, P, L.PDEC_SAVE, **f32),
               loss_part=torch.empty(n_nets, tiles, **f32), loss=torch.empty(n_nets, **f32))
    a.pred, a.saved, a.loss_part, a.loss = (out[k].data_ptr() for k in ("pred", "saved", "loss_part", "loss"))
    lib.call("iplan_pdec_fwd", a, L.current_stream(dev))
    out["_args"] = a
    out["_keep"] = (x0, h0, target, mask, keep, teacher)
    return out


def pdec_backward(arena, fwd, lib=None):
    """Backward of pdec_forward's loss (d loss = 1 per net): fills arena.grad, returns dLoss/dh0 [n_nets, rows, 32]."""
    lib = _lib(lib)
    a = fwd["_args"]
    n_nets, rows, P, d = a.n_nets, a.rows, a.P, a.d
    h0 = fwd["_keep"][1]
    dev = h0.device
    dsave = torch.empty(n_nets, rows, P, L.PDEC_DSAVE, dtype=torch.float32, device=dev)
    g_h0 = torch.empty(n_nets, rows, 32, dtype=torch.float32, device=dev)
    a.dsave, a.g_h0 = dsave.data_ptr(), g_h0.data_ptr()
    lib.call("iplan_pdec_bwd", a, L.current_stream(dev))
    SV, DS, H = L.PDEC_SAVE, L.PDEC_DSAVE, 32
    sv, dp = fwd["saved"].data_ptr(), dsave.data_ptr()
    sst, dst = (rows * P * SV, P * SV, SV), (rows * P * DS, P * DS, DS)
    off = arena.off
    w = Wgrad(arena.grad, n_nets)
    w.add(dp, dst, d, rows, P, x=sv + 4 * 208, x_strides=sst, K=H, dw_off=off("decoder.out.weight"), db_off=off("decoder.out.bias"))
    w.add(dp + 4 * 48, dst, 3 * H, rows, P, x=sv + 4 * 16, x_strides=sst, K=H,
          dw_off=off("decoder.rnn.weight_ih_l0"), db_off=off("decoder.rnn.bias_ih_l0"))
    w.add(dp + 4 * 48, dst, 3 * H, rows, P, x=sv + 4 * 176, x_strides=sst, K=H, x_shift=-1, x0=h0, x0_strides=(rows * H, H),
          dw_off=off("decoder.rnn.weight_hh_l0"), db_off=off("decoder.rnn.bias_hh_l0"), seg=(2 * H, 0, 3 * H))
    w.add(dp + 4 * 16, dst, H, rows, P, x=sv, x_strides=sst, K=d, dw_off=off("decoder.linear.weight"), db_off=off("decoder.linear.bias"))
    w._keep += [dsave, fwd]
    w.run(lib)
    return g_h0


# ---- behaviour learning ---------------------------------------------------------------------------------
def _beh_pieces(which, default):
    """Window pieces of the behaviour forward / backward pipelines (IPLAN_BEH_PIECES[_FWD|_BWD]: tuning / test knobs)."""
    return int(os.environ.get("IPLAN_BEH_PIECES_" + which, os.environ.get("IPLAN_BEH_PIECES", str(default))))


def _piece_bounds(J, pieces):
    """Window ranges of the pipelined behaviour forward / BPTT: ``pieces`` equal ranges behind ONE short range at the low end
    (windows [0, J // 16)).  The forward walks the ranges bottom up and its decoder can only start once the encoder has
    finished the first range; the BPTT walks them top down and the encoder's BPTT of the last range -- plus the gradient
    reduction and the optimiser step the next rollout waits for -- trails the decoder's.  With equal ranges both waits were a
    whole range of encoder work (0.65 ms in front of the decoder forward, 1.3 ms behind the decoder BPTT: rocprofv3 trace of
    a training cycle, profiles/history/r02h_cycle_trace_learn_phase.txt); the short range makes them a sixteenth of the episode."""
    if pieces <= 1 or J < 4 * pieces or os.environ.get("IPLAN_BEH_EQUAL_PIECES"):
        return [round(J * k / pieces) for k in range(pieces + 1)]
    s0 = max(1, J // 16)
    return [0] + [s0 + round((J - s0) * k / pieces) for k in range(pieces + 1)]


def _beh_thin_in_kernel():
    """decoder BPTT second form with the thin weight gradients accumulated in the kernel (the default): not with the first form
    (IPLAN_DEC_BWD_V1=1) nor with the A/B knob IPLAN_DEC_THIN_ROWS=1"""
    return not os.environ.get("IPLAN_DEC_BWD_V1") and not os.environ.get("IPLAN_DEC_THIN_ROWS")


def beh_forward(enc_arena, dec_arena, hist, mask, L_win, Z, coef, thres, drop_p, keep=None, seed=0, hard=False, win_norm=None,
                lib=None):
    """Forward of Behavior_policy.learn for all nets.  hist [n_nets, E, T, N, d] (first three dims may be
    strided), mask [n_nets, E, T] contiguous, keep uint8 [n_nets, J, E*N, L, 64] or None (in-kernel draw from
    ``seed``).  Returns dict(loss [n_nets, 2] = (behaviour, stability), saved...)."""
    lib = _lib(lib)
    n_nets, E, T, N, d = hist.shape
    assert hist.dtype == torch.float32 and hist.stride(4) == 1 and hist.stride(3) == d
    assert mask.shape == (n_nets, E, T) and mask.is_contiguous() and mask.dtype == torch.float32
    dev = hist.device
    f32 = dict(dtype=torch.float32, device=dev)
    J = (T // L_win - 1) if hard else (T - 1 - L_win)
    rows = E * N
    a = L.BehArgs()
    a.n_nets, a.E, a.N, a.T, a.L, a.d, a.Z = n_nets, E, N, T, L_win, d, Z
    a.hard = 1 if hard else 0
    a.hist, a.h_s_net, a.h_s_e, a.h_s_t = hist.data_ptr(), hist.stride(0), hist.stride(1), hist.stride(2)
    a.mask = mask.data_ptr()
    if keep is not None:
        assert keep.dtype == torch.uint8 and keep.shape == (n_nets, J, rows, L_win, 64) and keep.is_contiguous()
        a.keep = keep.data_ptr()
    a.seed, a.drop_p, a.coef, a.thres = seed, drop_p, coef, thres
    # the output layer's input is only read by the out.weight contraction of iplan_wgrad; the BPTT's second form accumulates that
    # gradient itself (beh_backward: ``thin``) and the forward then need not store it (64 of 480 floats per chain-step)
    a.fwd_skip_act = 1 if (_beh_thin_in_kernel() and not os.environ.get("IPLAN_FWD_SAVE_ACT")) else 0     # (IPLAN_FWD_SAVE_ACT=1: A/B knob)
    if win_norm is not None:                               # per-window mask sums over all data-parallel ranks' envs
        assert win_norm.shape == (n_nets, J) and win_norm.dtype == torch.float32 and win_norm.is_contiguous()
        a.win_norm = win_norm.data_ptr()
    a.enc_params, a.enc_s_net = enc_arena.data.data_ptr(), enc_arena.net_stride
    for i, k in enumerate(L.ENC_PARAM_ORDER):
        a.enc_off[i] = enc_arena.off(k)
    a.dec_params, a.dec_s_net = dec_arena.data.data_ptr(), dec_arena.net_stride
    for i, k in enumerate(L.DEC_PARAM_ORDER):
        a.dec_off[i] = dec_arena.off(k)
    tiles = (rows + 15) // 16
    # decoder record, column-grouped: [net, chain tile, 16-column group (31), step (J * L), chain (16), 16] -- a wave's store of one
    # 16-column group of its 16 chains is ONE contiguous 1 KiB block (5.6 instead of 3.3 TB/s of stores, scripts/ubench/
    # record_store.hip), and a column group's rows are contiguous over (step, chain) for the weight-gradient contraction.  The chain
    # slots of a ragged last tile are never written by the kernels: zeroed here, they add nothing to the gradients.
    out = dict(saved_dec=torch.empty(n_nets, tiles, L.BEH_SAVE_DEC // 16, J * L_win, 16, 16, **f32),
               saved_enc=torch.empty(n_nets, tiles, L.BEH_SAVE_ENC // 16, J * L_win, 16, 16, **f32),     # column-grouped too
               saved_lat=torch.empty(n_nets, rows, J, L.BEH_SAVE_LAT, **f32),
               loss_part=torch.empty(n_nets, tiles, 2, **f32), loss=torch.empty(n_nets, 2, **f32))
    if rows % 16:
        out["saved_dec"][:, -1, :, :, rows % 16:].zero_()
    for k in ("saved_dec", "saved_enc", "saved_lat", "loss_part", "loss"):
        setattr(a, k, out[k].data_ptr())
    # Pipeline (GPU): the decoder holds 138 of the 256 CUs for the whole episode, so the forward runs in window pieces --
    # the encoder of piece k+1 on a side stream beside the decoder of piece k on the main stream.
    side = None
    if dev.type == "cuda" and not os.environ.get("IPLAN_BEH_SERIAL"):
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev, main.cuda_stream)
    pieces = max(1, min(_beh_pieces("FWD", 4 if side is not None else 1), J))
    stream = L.current_stream(dev)
    if pieces == 1:
        lib.call("iplan_beh_fwd", a, stream)
    else:
        bounds = _piece_bounds(J, pieces)
        pieces = len(bounds) - 1
        out["enc_carry"] = torch.empty(n_nets, tiles, 768, **f32)
        out["dec_carry"] = torch.empty(n_nets, tiles, 2, 512, **f32)
        a.enc_carry, a.dec_carry = out["enc_carry"].data_ptr(), out["dec_carry"].data_ptr()
        if side is not None:
            side.wait_stream(main)                           # the inputs were produced on the main stream
        enc_done = []
        for k in range(pieces):                              # encoder pieces, back to back (side stream on the GPU)
            a.fwd_phase, a.fwd_j_lo, a.fwd_j_hi = 1, bounds[k], bounds[k + 1]
            if side is None:
                lib.call("iplan_beh_fwd", a, stream)
            else:
                _launch("beh_enc_fwd_kernel", lambda: lib.call("iplan_beh_fwd", a, side.cuda_stream), stream=side)
                ev = torch.cuda.Event()
                ev.record(side)
                enc_done.append(ev)
        for k in range(pieces):                              # decoder pieces, each behind its encoder piece
            if side is not None:
                main.wait_event(enc_done[k])
            a.fwd_phase, a.fwd_j_lo, a.fwd_j_hi = 2, bounds[k], bounds[k + 1]
            _launch("beh_dec_fwd_kernel", lambda: lib.call("iplan_beh_fwd", a, stream))
        a.fwd_phase, a.fwd_j_lo, a.fwd_j_hi = 3, 0, 0
        lib.call("iplan_beh_fwd", a, stream)
        a.fwd_phase = 0
        if side is not None:
            for k in ("saved_enc", "saved_lat", "enc_carry"):
                out[k].record_stream(side)
            for t in (hist, mask):
                t.record_stream(side)
    out["_args"] = a
    out["_keep"] = (hist, mask, keep, win_norm)
    return out


def beh_window_mask_sums(mask, L_win, hard=False):
    """[n_nets, J] float32: for every window the sum of ``mask`` [n_nets, E, T] over the envs and the window's target
    steps -- what the behaviour kernels sum in-kernel when no ``win_norm`` is given (sums of 0/1 flags: exact)."""
    n_nets, E, T = mask.shape
    col = mask.sum(dim=1)                                  # [n_nets, T]
    if hard:
        J = T // L_win - 1
        return col[:, :J * L_win].sum(dim=1, keepdim=True).expand(n_nets, J).contiguous()
    J = T - 1 - L_win
    cs = torch.cat([torch.zeros(n_nets, 1, dtype=col.dtype, device=col.device), col.cumsum(dim=1)], dim=1)   # cs[k] = sum_{t<k}
    j = torch.arange(J, device=col.device)
    return (cs[:, j + 1 + L_win] - cs[:, j + 1]).contiguous()


def beh_backward(enc_arena, dec_arena, fwd, accumulate=False, penalty=0.0, E_norm=0, defer_dec_wgrad=False, lib=None):
    """BPTT of beh_forward's behaviour loss: fills both gradient arenas (``accumulate=True``: adds to them -- launches on
    disjoint env chunks of one batch, normalised by a shared ``win_norm``).
    ``defer_dec_wgrad=True``: the DECODER's weight-gradient contraction is not launched; the returned dict carries
    ``"dec_wgrad"``, a callable ``(stream)`` that enqueues it (one call over all steps) on ``stream`` behind the BPTT.  The
    caller decides where that work runs -- nothing but the decoder's own optimiser step depends on it."""
    lib = _lib(lib)
    a = fwd["_args"]
    n_nets, E, N, T, Lw, d, Z = a.n_nets, a.E, a.N, a.T, a.L, a.d, a.Z
    J, rows = ((T // Lw - 1) if a.hard else (T - 1 - Lw)), E * N
    dev = fwd["loss"].device
    f32 = dict(dtype=torch.float32, device=dev)
    tiles = (rows + 15) // 16
    dd = torch.empty(n_nets, tiles, L.BEH_DSAVE_DEC // 16, J * Lw, 16, 16, **f32)          # column-grouped like saved_dec (beh_forward)
    if rows % 16:
        dd[:, -1, :, :, rows % 16:].zero_()
    dl = torch.empty(n_nets, rows, J, L.BEH_DSAVE_LAT, **f32)
    ep = torch.empty(n_nets, tiles, L.BEH_ENC_PART, **f32)
    a.dsave_dec, a.dsave_lat, a.enc_part = dd.data_ptr(), dl.data_ptr(), ep.data_ptr()
    a.enc_grad, a.enc_grad_s_net = enc_arena.grad.data_ptr(), enc_arena.grad.stride(0)
    a.enc_grad_beta = 1.0 if accumulate else 0.0
    a.penalty, a.E_norm = float(penalty), int(E_norm)      # behavior_variation_penalty: the stability term's weight in the loss
    # decoder BPTT, second form (the default): the thin weight gradients (out.*, linear.*) are accumulated in the kernel and
    # reduced into the arena by its last window range; the first form (IPLAN_DEC_BWD_V1=1) and IPLAN_DEC_THIN_ROWS=1 (A/B knob)
    # stream row gradients to iplan_wgrad instead
    thin = _beh_thin_in_kernel()
    assert thin or not a.fwd_skip_act, "the forward skipped the output layer's input: the row-gradient path needs it"
    tp = None
    if thin:
        tp = torch.empty(n_nets, (tiles + L.BEH_DEC_BWD2_TILES - 1) // L.BEH_DEC_BWD2_TILES, L.BEH_DEC_THIN_PART, **f32)
        a.dec_thin_part, a.dec_grad, a.dec_grad_s_net = tp.data_ptr(), dec_arena.grad.data_ptr(), dec_arena.grad.stride(0)
        a.dec_grad_beta = 1.0 if accumulate else 0.0
    else:
        a.dec_thin_part = None
    SD, DD = L.BEH_SAVE_DEC, L.BEH_DSAVE_DEC
    n_in = J * Lw
    sd = fwd["saved_dec"].data_ptr()
    # rows of the contraction = (chain tile, step * 16 + chain): 16 floats apart inside a column group, groups n_in * 256 apart
    cgs = n_in * 256
    sd_st, dd_st = (tiles * n_in * 16 * SD, n_in * 16 * SD, 16), (tiles * n_in * 16 * DD, n_in * 16 * DD, 16)
    H = 64
    off = dec_arena.off

    def dec_wgrad(s0, s1, beta):
        """decoder weight gradients over the rows of steps [s0, s1) of every chain (accumulating when beta = 1)"""
        w = Wgrad(dec_arena.grad, n_nets, tag="iplan_wgrad:beh_dec")
        ddp, sdp, n = dd.data_ptr() + 4 * s0 * 256, sd + 4 * s0 * 256, (s1 - s0) * 16
        cg = dict(dy_cg_stride=cgs, x_cg_stride=cgs)
        if not thin:
            w.add(ddp, dd_st, d, tiles, n, x=sdp, x_strides=sd_st, K=H, x_col0=416, beta=beta,
                  dw_off=off("decoder.out.weight"), db_off=off("decoder.out.bias"), **cg)
        w.add(ddp, dd_st, 3 * H, tiles, n, x=sdp, x_strides=sd_st, K=H, x_col0=32, beta=beta, seg=(3 * H, 80, 0),
              dw_off=off("decoder.rnn.weight_ih_l0"), db_off=off("decoder.rnn.bias_ih_l0"), **cg)
        # recurrent operand = the previous step's hidden state (16 rows back); the step before a range's first one is still in
        # the record and is read in place, the step before step 0 is zero
        w.add(ddp, dd_st, 3 * H, tiles, n, x=sdp, x_strides=sd_st, K=H, x_col0=352, x_shift=-16, x_pre_valid=s0 > 0, beta=beta,
              dw_off=off("decoder.rnn.weight_hh_l0"), db_off=off("decoder.rnn.bias_hh_l0"), seg=(2 * H, 80, 80 + 3 * H), **cg)
        # input Linear: the record keeps its input row [x_t || latent] as one tile
        if not thin:
            w.add(ddp, dd_st, H, tiles, n, x=sdp, x_strides=sd_st, K=d + Z, beta=beta, seg=(H, 16, 0),
                  dw_off=off("decoder.linear.weight"), db_off=off("decoder.linear.bias"), **cg)
        w._keep += [dd, fwd, tp]
        w.run(lib)

    # Pipeline: the decoder BPTT runs in pieces (top windows first) on the main stream; the weight-gradient contraction
    # of the rows a piece produced and the encoder's BPTT over the same windows (it needs their d(loss)/d(latent)) run
    # on two side streams beside the next decoder piece (the decoder kernel holds 138 of the 256 CUs).
    side = side2 = None
    if dev.type == "cuda" and not os.environ.get("IPLAN_BEH_SERIAL"):       # (the knob is for kernel timing experiments)
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev, main.cuda_stream)
        side2 = _side_stream(dev, main.cuda_stream, 2)
    if defer_dec_wgrad and side is None and dev.type == "cuda":
        main = torch.cuda.current_stream(dev)
    pieces = max(1, min(_beh_pieces("BWD", 6 if side is not None else 1), J))
    bounds = _piece_bounds(J, pieces)
    if thin:
        # the in-kernel thin gradients exist in the BPTT's second form only, which takes at most BEH_D2_MAX_WINDOWS windows per launch
        # (iplan_beh_bwd returns EINVAL beyond): episodes longer than that get more pieces instead of failing
        while max(b1 - b0 for b0, b1 in zip(bounds[:-1], bounds[1:])) > L.BEH_D2_MAX_WINDOWS:
            pieces += 1
            bounds = [round(J * k / pieces) for k in range(pieces + 1)]
    pieces = len(bounds) - 1
    carry = torch.empty(n_nets, tiles, 2, 512, **f32)
    ecarry = torch.empty(n_nets, tiles, 768, **f32)
    a.dec_carry, a.enc_carry = carry.data_ptr(), ecarry.data_ptr()
    stream = L.current_stream(dev)
    for k in range(pieces, 0, -1):
        a.bwd_phase, a.bwd_j_lo, a.bwd_j_hi = 1, bounds[k - 1], bounds[k]
        _launch("beh_dec_bwd_kernel", lambda: lib.call("iplan_beh_bwd", a, stream))
        beta = 0.0 if (k == pieces and not accumulate) else 1.0
        if side is None:
            if not defer_dec_wgrad:
                dec_wgrad(bounds[k - 1] * Lw, bounds[k] * Lw, beta)
            a.bwd_phase = 2
            lib.call("iplan_beh_bwd", a, stream)
        else:
            ev = torch.cuda.Event()
            ev.record(main)
            side2.wait_event(ev)
            if not defer_dec_wgrad:
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    dec_wgrad(bounds[k - 1] * Lw, bounds[k] * Lw, beta)
            a.bwd_phase = 2
            _launch("beh_enc_bwd_kernel", lambda: lib.call("iplan_beh_bwd", a, side2.cuda_stream), stream=side2)
    a.bwd_phase, a.bwd_j_lo, a.bwd_j_hi = 0, 0, 0
    out = dict(dsave_dec=dd, dsave_lat=dl, enc_part=ep, dec_thin_part=tp)
    if defer_dec_wgrad:
        ev_bptt = None
        if dev.type == "cuda":
            ev_bptt = torch.cuda.Event()
            ev_bptt.record(main)                                         # every decoder BPTT range is behind this

        def run_dec_wgrad(strm=None):
            if strm is None or dev.type != "cuda":
                dec_wgrad(0, J * Lw, 1.0 if accumulate else 0.0)
                return
            strm.wait_event(ev_bptt)
            # ... and behind whatever the caller has enqueued on the main stream by now (Behavior_policy.learn: the encoder's
            # BPTT tail, gradient reduction and optimiser step -- what the next rollout waits for).  The wide contraction's
            # 372-register waves take whole CUs: started right behind the last decoder range it ran beside the encoder's last
            # range and stretched it from 0.6 to 3.9 ms (rocprofv3 trace, profiles/r03c_notes.md).
            ev_now = torch.cuda.Event()
            ev_now.record(torch.cuda.current_stream(dev))
            strm.wait_event(ev_now)
            with torch.cuda.stream(strm):
                dec_wgrad(0, J * Lw, 1.0 if accumulate else 0.0)
            for t in (dd, fwd["saved_dec"]):
                t.record_stream(strm)
        out["dec_wgrad"] = run_dec_wgrad
    if side is not None:
        for st in ((side2,) if defer_dec_wgrad else (side, side2)):
            ev_done = torch.cuda.Event()
            ev_done.record(st)
            main.wait_event(ev_done)
        if not defer_dec_wgrad:
            for t in (dd, carry, fwd["saved_dec"]):                     # touched by the side streams: keep the allocator honest
                t.record_stream(side)
        for t in (dl, ep, ecarry, fwd["saved_enc"], fwd["saved_lat"]):
            t.record_stream(side2)
    return out


def bdec_forward(enc_arena, dec_arena, window, latent, hidden, drop_p=0.0, keep=None, seed=0, lib=None):
    """Behavior_Latent_Decoder.forward on one explicit window for all nets (single-window mode of iplan_beh_fwd).
    window [n_nets, rows, L, d], latent [n_nets, rows, Z], hidden [n_nets, rows, 64] -> (pred [n_nets, rows, L, d],
    new hidden [n_nets, rows, 64])."""
    lib = _lib(lib)
    n_nets, rows, Lw, d = window.shape
    Z = latent.shape[-1]
    dev = window.device
    f32 = dict(dtype=torch.float32, device=dev)
    for t in (window, latent, hidden):
        assert t.is_contiguous() and t.dtype == torch.float32
    a = L.BehArgs()
    a.n_nets, a.E, a.N, a.T, a.L, a.d, a.Z = n_nets, rows, 1, Lw + 2, Lw, d, Z
    a.win, a.lat_in, a.hd_in = window.data_ptr(), latent.data_ptr(), hidden.data_ptr()
    pred, hout = torch.empty(n_nets, rows, Lw, d, **f32), torch.empty(n_nets, rows, 64, **f32)
    saved = torch.empty(n_nets, (rows + 15) // 16 * 16, 1, Lw, L.BEH_SAVE_DEC, **f32)        # (scratch record: column-grouped by chain tile)
    a.pred_out, a.hd_out, a.saved_dec = pred.data_ptr(), hout.data_ptr(), saved.data_ptr()
    if keep is not None:
        assert keep.dtype == torch.uint8 and keep.shape == (n_nets, 1, rows, Lw, 64) and keep.is_contiguous()
        a.keep = keep.data_ptr()
    a.seed, a.drop_p, a.coef, a.thres = seed, drop_p, 0.0, 0.0
    a.enc_params, a.enc_s_net = enc_arena.data.data_ptr(), enc_arena.net_stride
    for i, k in enumerate(L.ENC_PARAM_ORDER):
        a.enc_off[i] = enc_arena.off(k)
    a.dec_params, a.dec_s_net = dec_arena.data.data_ptr(), dec_arena.net_stride
    for i, k in enumerate(L.DEC_PARAM_ORDER):
        a.dec_off[i] = dec_arena.off(k)
    lib.call("iplan_beh_fwd", a, L.current_stream(dev))
    return pred, hout


# ---- FC behaviour ablation: stacked three-layer perceptrons -------------------------------------------------
MLP3_PARAM_ORDER = ("linear_1.weight", "linear_1.bias", "linear_2.weight", "linear_2.bias", "out.weight", "out.bias")


def mlp3_forward(arena, prefix, x, H, O, softmax=False, target=None, save=True, lib=None):
    """out = [softmax](W3 tanh(W2 tanh(W1 x + b1) + b2) + b3) for all nets: x [n_nets, rows, K0] -> out [n_nets, rows, O].
    ``prefix``: parameter-name prefix inside the arena ("" or "decoder.").  With ``target`` also returns the per-net sum of
    |target - out| (L1 numerator).  Returns dict(out, saved, l1, _args)."""
    lib = _lib(lib)
    n_nets, rows, K0 = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32
    dev = x.device
    f32 = dict(dtype=torch.float32, device=dev)
    a = L.Mlp3Args()
    a.n_nets, a.K0, a.H, a.O, a.softmax, a.rows = n_nets, K0, H, O, 1 if softmax else 0, rows
    a.x, a.params, a.params_s_net = x.data_ptr(), arena.data.data_ptr(), arena.net_stride
    for i, k in enumerate(MLP3_PARAM_ORDER):
        a.off[i] = arena.off(prefix + k)
    out = dict(out=torch.empty(n_nets, rows, O, **f32))
    a.out = out["out"].data_ptr()
    if save:
        out["saved"] = torch.empty(n_nets, rows, 2 * H, **f32)
        a.saved = out["saved"].data_ptr()
    if target is not None:
        assert target.shape == (n_nets, rows, O) and target.is_contiguous()
        part = torch.empty(n_nets, 4 * ((rows + 63) // 64), **f32)
        a.target, a.loss_part = target.data_ptr(), part.data_ptr()
    lib.call("iplan_mlp3_fwd", a, L.current_stream(dev))
    if target is not None:
        out["l1"] = part.sum(dim=1)
    out["_args"] = a
    out["_keep"] = (x, target)
    return out


def mlp3_backward(arena, prefix, fwd, g_out=None, g_scale=0.0, want_dx=False, beta=0.0, lib=None):
    """Backward of mlp3_forward: fills the arena's gradient entries of the six tensors (accumulating when beta = 1) and
    returns dx [n_nets, rows, K0] if asked.  d(out) = g_out, or the L1 loss's -sign(target - out) * g_scale."""
    lib = _lib(lib)
    a = fwd["_args"]
    n_nets, rows, K0, H, O = a.n_nets, a.rows, a.K0, a.H, a.O
    dev = fwd["out"].device
    f32 = dict(dtype=torch.float32, device=dev)
    O16 = 16 * ((O + 15) // 16)
    DS = 2 * H + O16
    ds = torch.empty(n_nets, rows, DS, **f32)
    a.dsave = ds.data_ptr()
    dx = torch.empty(n_nets, rows, K0, **f32) if want_dx else None
    a.dx = dx.data_ptr() if want_dx else None
    if g_out is not None:
        assert g_out.shape == (n_nets, rows, O) and g_out.is_contiguous()
        a.g_out = g_out.data_ptr()
    else:
        a.g_out, a.g_scale = None, g_scale
    lib.call("iplan_mlp3_bwd", a, L.current_stream(dev))
    off = arena.off
    sv, x = fwd["saved"], fwd["_keep"][0]
    w = Wgrad(arena.grad, n_nets)
    dst = (rows * DS, DS, DS)
    w.add(ds.data_ptr(), dst, H, rows, 1, x=x.data_ptr(), x_strides=(rows * K0, K0, K0), K=K0, beta=beta,
          dw_off=off(prefix + "linear_1.weight"), db_off=off(prefix + "linear_1.bias"))
    w.add(ds.data_ptr() + 4 * H, dst, H, rows, 1, x=sv.data_ptr(), x_strides=(rows * 2 * H, 2 * H, 2 * H), K=H, beta=beta,
          dw_off=off(prefix + "linear_2.weight"), db_off=off(prefix + "linear_2.bias"))
    w.add(ds.data_ptr() + 8 * H, dst, O, rows, 1, x=sv.data_ptr() + 4 * H, x_strides=(rows * 2 * H, 2 * H, 2 * H), K=H, beta=beta,
          dw_off=off(prefix + "out.weight"), db_off=off(prefix + "out.bias"))
    w._keep += [ds, sv, x]
    w.run(lib)
    return dx
