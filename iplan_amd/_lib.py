"""ctypes binding of the C ABI declared in include/iplan_hip.h.

``get_lib()`` loads the gfx950 build (``iplan_amd/libiplan_hip.so``) and raises if it is missing:
there is no CPU or PyTorch fallback anywhere in the product path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IPLAN_HIP_LIB", os.path.join(_HERE, "libiplan_hip.so"))   # override = kernel A/B experiments

GAT_NPARAM = 20
GAT_PARAM_ORDER = [
    "encoding.weight", "encoding.bias",
    "hard_bi_GRU.weight_ih_l0", "hard_bi_GRU.weight_hh_l0", "hard_bi_GRU.bias_ih_l0", "hard_bi_GRU.bias_hh_l0",
    "hard_bi_GRU.weight_ih_l0_reverse", "hard_bi_GRU.weight_hh_l0_reverse",
    "hard_bi_GRU.bias_ih_l0_reverse", "hard_bi_GRU.bias_hh_l0_reverse",
    "hard_encoding.weight", "hard_encoding.bias",
    "q.weight", "k.weight", "v.weight", "v.bias",
    "rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh",
]

fp = C.c_void_p
i32 = C.c_int32
i64 = C.c_int64


class GatSaved(C.Structure):
    _fields_ = [(k, fp) for k in ("h_enc", "gru", "qkv", "soft", "hard", "x", "cell")]


class GatFwdArgs(C.Structure):
    _fields_ = [
        ("n_nets", i32), ("B", i32), ("N", i32), ("d0", i32), ("d1", i32),
        ("src0", fp), ("src0_s_net", i64), ("src0_s_b", i64),
        ("src1", fp), ("src1_s_net", i64), ("src1_s_b", i64),
        ("h_prev", fp), ("h_s_net", i64), ("h_s_b", i64),
        ("out", fp), ("out_s_net", i64), ("out_s_b", i64),
        ("noise", fp), ("params", fp), ("params_s_net", i64),
        ("off", i64 * GAT_NPARAM), ("tau", C.c_float), ("saved", GatSaved), ("phase_clocks", fp),
    ]


class IplanError(RuntimeError):
    pass


# every entry point include/iplan_hip.h declares
ENTRY_POINTS = ["iplan_gat_fwd", "iplan_enc_fwd", "iplan_ac_fwd", "iplan_adam_step", "iplan_wgrad",
                "iplan_ac_bwd_tail", "iplan_ac_bwd_fc1", "iplan_ac_bwd_fc1_finalize", "iplan_ppo_prepare", "iplan_ppo_adv_norm", "iplan_ppo_loss", "iplan_gat_bwd",
                "iplan_pdec_fwd", "iplan_pdec_bwd", "iplan_beh_fwd", "iplan_beh_bwd", "iplan_mlp3_fwd", "iplan_mlp3_bwd", "iplan_seq2seq_fwd", "iplan_ac_pack_fc1",
                "iplan_ac_xhat_pack", "iplan_ac_fc1_split_fwd", "iplan_ac_bwd_fc1_split",
                "iplan_p2p_publish", "iplan_p2p_reduce", "iplan_obs_history_step", "iplan_seq2seq_bwd"]
RAW_ENTRY_POINTS = ["iplan_grad_sqnorm", "iplan_wgrad_workspace_floats", "iplan_ac_kpad", "iplan_ac_fc1_groups", "iplan_sizeof", "iplan_ac_packed_floats",
                    "iplan_p2p_alloc", "iplan_p2p_free", "iplan_p2p_export", "iplan_p2p_open", "iplan_p2p_close", "iplan_gat_enc_fwd", "iplan_gat_enc_ac_fwd", "iplan_gumbel_noise", "iplan_ac_xhat_floats", "iplan_ac_fc1_split_chunks", "iplan_ac_fc1_split_parts"]      # non (args*, stream) signatures


class Lib:
    """Thin typed wrapper around a loaded libiplan_*.so."""

    def __init__(self, cdll):
        self.c = cdll
        cdll.iplan_last_error.restype = C.c_char_p
        cdll.iplan_version.restype = C.c_int
        for name in ENTRY_POINTS:
            if not hasattr(cdll, name):
                raise IplanError(f"{name} missing from the loaded library")
            fn = getattr(cdll, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p]
        for name in RAW_ENTRY_POINTS:
            if not hasattr(cdll, name):
                raise IplanError(f"{name} missing from the loaded library")
            getattr(cdll, name).restype = C.c_int
        cdll.iplan_sizeof.restype = C.c_size_t
        cdll.iplan_sizeof.argtypes = [C.c_char_p]
        cdll.iplan_ac_packed_floats.restype = C.c_int64
        cdll.iplan_ac_packed_floats.argtypes = [C.c_void_p]
        cdll.iplan_ac_fc1_split_chunks.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        cdll.iplan_ac_fc1_split_parts.argtypes = [C.c_int32, C.c_int32]
        cdll.iplan_ac_xhat_floats.restype = C.c_int64
        cdll.iplan_ac_xhat_floats.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        cdll.iplan_wgrad_workspace_floats.restype = C.c_size_t
        cdll.iplan_wgrad_workspace_floats.argtypes = [C.c_void_p]
        cdll.iplan_gat_enc_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        cdll.iplan_gat_enc_ac_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        cdll.iplan_gumbel_noise.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_void_p]
        cdll.iplan_p2p_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        cdll.iplan_p2p_free.argtypes = [C.c_void_p]
        cdll.iplan_p2p_export.argtypes = [C.c_void_p, C.POINTER(IpcHandle)]
        cdll.iplan_p2p_open.argtypes = [C.POINTER(IpcHandle), C.POINTER(C.c_void_p)]
        cdll.iplan_p2p_close.argtypes = [C.c_void_p]

    def call(self, name, args, stream=None):
        rc = getattr(self.c, name)(C.byref(args), C.c_void_p(stream or 0))
        if rc != 0:
            raise IplanError(f"{name} failed ({rc}): {self.c.iplan_last_error().decode()}")


_lib = None
_test_override = None


def use_library_for_tests(lib):
    """Test seam ONLY: the CPU test-suite injects the host-emulated build of the same kernel
    sources (tests/emu).  Product code never calls this; get_lib() itself only ever loads the
    gfx950 build and raises when it is missing."""
    global _test_override
    _test_override = lib


def get_lib():
    global _lib
    if _test_override is not None:
        return _test_override
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IplanError(
                f"{LIB_PATH} not found: build it with `make -C iplan_amd/csrc` (or __graft_entry__.build()). "
                "iplan_amd has no CPU fallback.")
        _lib = Lib(C.CDLL(LIB_PATH))
    return _lib


def current_stream(device):
    """Raw hipStream_t of torch's current stream on `device` (0 = null stream on CPU/emulation)."""
    if torch.device(device).type == "cuda":
        return torch.cuda.current_stream(device).cuda_stream
    return 0


def ptr(t):
    return None if t is None else t.data_ptr()


# ---- encoder -------------------------------------------------------------------------------------
ENC_PARAM_ORDER = ["linear.weight", "linear.bias", "rnn.weight_ih_l0", "rnn.weight_hh_l0",
                   "rnn.bias_ih_l0", "rnn.bias_hh_l0", "out.weight", "out.bias"]


class EncFwdArgs(C.Structure):
    _fields_ = [
        ("n_nets", i32), ("B", i32), ("N", i32), ("L", i32), ("d", i32), ("Z", i32),
        ("x", fp), ("x_s_net", i64), ("x_s_b", i64),
        ("h0", fp), ("h0_s_net", i64), ("h0_s_b", i64),
        ("hL", fp), ("hL_s_net", i64), ("hL_s_b", i64),
        ("prev_latent", fp), ("pl_s_net", i64), ("pl_s_b", i64),
        ("latent_out", fp), ("lo_s_net", i64), ("lo_s_b", i64),
        ("one_minus_c", C.c_float), ("c", C.c_float),
        ("params", fp), ("params_s_net", i64), ("off", i64 * len(ENC_PARAM_ORDER)),
        ("x_s_i", i64), ("x_s_t", i64),
    ]


# ---- actor / critic --------------------------------------------------------------------------------
AC_TRUNK_ORDER = [
    "base.feature_norm.weight", "base.feature_norm.bias",
    "base.mlp.fc1.0.weight", "base.mlp.fc1.0.bias", "base.mlp.fc1.2.weight", "base.mlp.fc1.2.bias",
    "base.mlp.fc2.0.0.weight", "base.mlp.fc2.0.0.bias", "base.mlp.fc2.0.2.weight", "base.mlp.fc2.0.2.bias",
    "rnn.rnn.weight_ih_l0", "rnn.rnn.weight_hh_l0", "rnn.rnn.bias_ih_l0", "rnn.rnn.bias_hh_l0",
    "rnn.norm.weight", "rnn.norm.bias",
]
ACTOR_PARAM_ORDER = AC_TRUNK_ORDER + ["act.action_out.linear.weight", "act.action_out.linear.bias"]
CRITIC_PARAM_ORDER = AC_TRUNK_ORDER + ["v_out.weight", "v_out.bias"]
AC_NPARAM = 18
AC_HIDDEN = 64
AC_SAVE_FLOATS = 10 * AC_HIDDEN + 8
AC_KS_SLOT_FLOATS = 16 * AC_HIDDEN + 32


class AcNet(C.Structure):
    _fields_ = [("params", fp), ("params_s_net", i64), ("off", i64 * AC_NPARAM), ("n_out", i32)]


class AcFeatures(C.Structure):
    _fields_ = [
        ("N", i32), ("w", i32 * 3), ("src", fp * 3), ("s_net", i64 * 3), ("s_row", i64 * 3),
        ("n_actions", i32), ("last_action", fp), ("la_s_net", i64), ("la_s_row", i64),
        ("n_id", i32), ("T", i32), ("T_phys", i32),
        ("last_action64", fp), ("la64_s_net", i64), ("la64_s_row", i64),
    ]


class AcFwdArgs(C.Structure):
    _fields_ = [
        ("n_agents", i32), ("rows", i32), ("which", i32), ("ksplit", i32),
        ("feat", AcFeatures), ("actor", AcNet), ("critic", AcNet),
        ("h_actor", fp), ("h_critic", fp), ("hs_net", i64), ("hs_row", i64),
        ("h_actor_out", fp), ("h_critic_out", fp),
        ("avail", fp), ("av_s_net", i64), ("av_s_row", i64),
        ("mode", i32), ("q_noise", fp),
        ("actions_in", fp), ("act_s_net", i64), ("act_s_row", i64),
        ("actions_out", fp), ("logp", fp), ("entropy", fp), ("probs", fp),
        ("values", fp), ("saved", fp),
        ("ho_s_net", i64), ("ho_s_row", i64), ("ao_s_net", i64), ("ao_s_row", i64),
        ("onehot_out", fp), ("oh_s_net", i64), ("oh_s_row", i64),
        ("ln_stats", fp), ("ln_stats_s_net", i64), ("ln_stats_mode", i32), ("phase_clocks", fp),
        ("packed_actor", fp), ("packed_critic", fp), ("packed_s_net", i64), ("fc1_pre", fp),
        ("ksplit_wg", i32), ("ks_scratch", fp), ("ks_count", fp), ("act_tanh", i32), ("fc1_pre_parts", i32),
    ]


class AcXhatArgs(C.Structure):
    _fields_ = [("n_agents", i32), ("rows", i32), ("feat", AcFeatures), ("ln_stats", fp), ("ln_stats_s_net", i64), ("xf", fp), ("xb", fp)]


class AcFc1SplitArgs(C.Structure):
    _fields_ = [("n_agents", i32), ("rows", i32), ("feat", AcFeatures), ("actor", AcNet), ("critic", AcNet), ("xf", fp),
                ("wsplit", fp), ("wbeta", fp), ("z1", fp), ("kparts", i32)]


class AcPackArgs(C.Structure):
    _fields_ = [("n_nets", i32), ("feat", AcFeatures), ("params", fp), ("params_s_net", i64), ("off_w1", i64), ("off_fn_w", i64),
                ("off_fn_b", i64), ("packed", fp), ("packed_s_net", i64), ("parts", i32)]


# ---- one-shot peer-to-peer all-reduce -----------------------------------------------------------------
P2P_MAX_RANKS = 8


class IpcHandle(C.Structure):
    _fields_ = [("bytes", C.c_ubyte * 64)]


class P2pArgs(C.Structure):
    _fields_ = [("world", i32), ("rank", i32), ("count", i64), ("data", fp), ("stage", fp * P2P_MAX_RANKS), ("flags", fp * P2P_MAX_RANKS),
                ("capacity", i64), ("seq", C.c_uint32), ("error", fp), ("spin_limit", i64)]


# ---- optimiser -------------------------------------------------------------------------------------
MAX_NETS = 16


class AdamArgs(C.Structure):
    _fields_ = [
        ("param", fp), ("grad", fp), ("exp_avg", fp), ("exp_avg_sq", fp),
        ("stride", i64), ("off", i64), ("n", i64), ("n_nets", i32),
        ("sqnorm", fp), ("sqnorm_stride", i32), ("sqnorm_slot", i32),
        ("max_norm", C.c_float), ("write_clipped", i32),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("bc1", C.c_float * MAX_NETS), ("bc2_sqrt", C.c_float * MAX_NETS), ("weight_decay", C.c_float),
    ]


# ---- weight-gradient contraction ---------------------------------------------------------------------
WGRAD_MAX = 16


class WgradProblem(C.Structure):
    _fields_ = [
        ("dy", fp), ("dy_s_net", i64), ("dy_s_outer", i64), ("dy_s_inner", i64),
        ("x", fp), ("x_s_net", i64), ("x_s_outer", i64), ("x_s_inner", i64),
        ("x0", fp), ("x0_s_net", i64), ("x0_s_outer", i64),
        ("dw_off", i64), ("db_off", i64), ("ws_off", i64),
        ("O", i32), ("K", i32), ("seg_split", i32), ("seg_c0", i32), ("seg_c1", i32),
        ("x_col0", i32), ("x_shift", i32), ("n_outer", i32), ("n_inner", i32),
        ("dw_ld", i32), ("dw_col0", i32), ("beta", C.c_float), ("scale", C.c_float),
        ("dy_cg_stride", i32), ("x_cg_stride", i32), ("x_pre_valid", i32),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("n_problems", i32), ("n_nets", i32), ("grad", fp), ("grad_s_net", i64),
        ("workspace", fp), ("workspace_floats", i64), ("p", WgradProblem * WGRAD_MAX),
    ]


# ---- actor / critic backward + PPO -------------------------------------------------------------------
AC_DSAVE_FLOATS = 6 * AC_HIDDEN + 16
AC_LNPART_FLOATS = 6 * AC_HIDDEN


class AcBwdArgs(C.Structure):
    _fields_ = [
        ("fwd", AcFwdArgs), ("g_logp", fp), ("g_entropy", fp), ("g_entropy_const", C.c_float),
        ("g_values", fp), ("dsave", fp), ("ln_part", fp), ("g_part", fp),
        ("fc1_chunk_rows", i32), ("fc1_chunks", i32), ("actor_grad", fp), ("critic_grad", fp),
        ("actor_grad_s_net", i64), ("critic_grad_s_net", i64), ("xb", fp),
    ]


class PpoPrepareArgs(C.Structure):
    _fields_ = [
        ("n_agents", i32), ("bs", i32), ("T", i32),
        ("reward", fp), ("rw_s_net", i64), ("rw_s_ep", i64), ("rw_s_t", i64),
        ("terminated", fp), ("tm_s_net", i64), ("tm_s_ep", i64), ("tm_s_t", i64),
        ("values", fp), ("gamma", C.c_float), ("lam", C.c_float),
        ("returns", fp), ("adv", fp), ("mask", fp), ("value_preds", fp), ("skip_norm", i32), ("no_gae", i32),
    ]


class AdvNormArgs(C.Structure):
    _fields_ = [("n_agents", i32), ("n", i32), ("row_stride", i64), ("adv", fp), ("sum", fp), ("sqdev", fp),
                ("count", C.c_float), ("phase", i32)]


class PpoLossArgs(C.Structure):
    _fields_ = [
        ("n_agents", i32), ("rows", i32), ("row_stride", i64),
        ("logp", fp), ("entropy", fp), ("values", fp), ("old_logp", fp), ("adv", fp),
        ("value_preds", fp), ("returns", fp), ("mask", fp),
        ("clip", C.c_float), ("huber_delta", C.c_float), ("value_loss_coef", C.c_float),
        ("g_logp", fp), ("g_values", fp), ("stats", fp), ("mask_sum", fp), ("flags", i32), ("row_count", C.c_float),
        ("n_parts", i32),
    ]


PPO_MSE, PPO_NO_VCLIP, PPO_VALUE_MEAN, PPO_POLICY_MEAN = 1, 2, 4, 8


# ---- GAT backward --------------------------------------------------------------------------------------
GAT_NODE_DY = 640
GAT_HARD_PART = 8 * 32 + 16
GAT_REC_GROUPS = 8                  # 16-column groups of the saved pair-GRU record per (ego tile, step): h r z n x 2 (csrc/gat.hip, gat_bwd.hip: REC = 8 * 256)
GAT_WHH_PART = 3 * 32 * 32 + 3 * 32


class GatBwdArgs(C.Structure):
    _fields_ = [("fwd", GatFwdArgs), ("g_out", fp), ("g_s_net", i64), ("g_s_b", i64),
                ("dgru", fp), ("node_dy", fp), ("hard_part", fp), ("whh_part", fp), ("grad", fp), ("grad_s_net", i64)]


# ---- prediction decoder ------------------------------------------------------------------------------
DEC_PARAM_ORDER = ["decoder.linear.weight", "decoder.linear.bias", "decoder.rnn.weight_ih_l0", "decoder.rnn.weight_hh_l0",
                   "decoder.rnn.bias_ih_l0", "decoder.rnn.bias_hh_l0", "decoder.out.weight", "decoder.out.bias"]
PDEC_SAVE, PDEC_DSAVE = 256, 176


class PdecArgs(C.Structure):
    _fields_ = [
        ("n_nets", i32), ("rows", i32), ("N", i32), ("P", i32), ("d", i32),
        ("x0", fp), ("h0", fp), ("target", fp), ("mask", fp), ("keep", fp), ("drop_p", C.c_float),
        ("teacher", fp), ("params", fp), ("params_s_net", i64), ("off", i64 * len(DEC_PARAM_ORDER)),
        ("pred", fp), ("saved", fp), ("loss_part", fp), ("loss", fp), ("dsave", fp), ("g_h0", fp), ("mask_sum", fp),
    ]


# ---- behaviour learning ------------------------------------------------------------------------------
BEH_SAVE_DEC, BEH_SAVE_ENC, BEH_SAVE_LAT = 496, 192, 32
BEH_ENC_PART = 7408
BEH_DSAVE_DEC, BEH_DSAVE_LAT = 336, 16
BEH_DEC_THIN_PART, BEH_DEC_BWD2_TILES = 2144, 3
BEH_D2_MAX_WINDOWS = 512                              # csrc/behavior_learn.hip: D2_MAX_WINDOWS (windows per launch of the decoder's second forms)


class BehArgs(C.Structure):
    _fields_ = [
        ("n_nets", i32), ("E", i32), ("N", i32), ("T", i32), ("L", i32), ("d", i32), ("Z", i32),
        ("hist", fp), ("h_s_net", i64), ("h_s_e", i64), ("h_s_t", i64),
        ("mask", fp), ("keep", fp), ("seed", C.c_uint64),
        ("drop_p", C.c_float), ("coef", C.c_float), ("thres", C.c_float),
        ("enc_params", fp), ("enc_s_net", i64), ("enc_off", i64 * len(ENC_PARAM_ORDER)),
        ("dec_params", fp), ("dec_s_net", i64), ("dec_off", i64 * len(DEC_PARAM_ORDER)),
        ("saved_dec", fp), ("saved_enc", fp), ("saved_lat", fp), ("loss_part", fp), ("loss", fp),
        ("dsave_dec", fp), ("dsave_lat", fp),
        ("win", fp), ("lat_in", fp), ("hd_in", fp), ("pred_out", fp), ("hd_out", fp), ("hard", i32),
        ("enc_part", fp), ("enc_grad", fp), ("enc_grad_s_net", i64), ("bwd_phase", i32),
        ("bwd_j_lo", i32), ("bwd_j_hi", i32), ("dec_carry", fp),
        ("fwd_phase", i32), ("fwd_j_lo", i32), ("fwd_j_hi", i32), ("enc_carry", fp), ("win_norm", fp),
        ("enc_grad_beta", C.c_float), ("penalty", C.c_float), ("E_norm", i32),
        ("dec_thin_part", fp), ("dec_grad", fp), ("dec_grad_s_net", i64), ("dec_grad_beta", C.c_float), ("fwd_skip_act", i32),
    ]


# ---- FC behaviour ablation -------------------------------------------------------------------------------
class Mlp3Args(C.Structure):
    _fields_ = [
        ("n_nets", i32), ("K0", i32), ("H", i32), ("O", i32), ("softmax", i32), ("rows", i64),
        ("x", fp), ("params", fp), ("params_s_net", i64), ("off", i64 * 6),
        ("out", fp), ("saved", fp), ("target", fp), ("loss_part", fp), ("g_out", fp), ("g_scale", C.c_float),
        ("dsave", fp), ("dx", fp),
    ]


class Seq2SeqArgs(C.Structure):
    _fields_ = [
        ("rows", i32), ("T_in", i32), ("In", i32), ("H", i32), ("layers", i32), ("P", i32), ("O", i32),
        ("x", fp), ("last", fp), ("teacher", fp), ("coins", fp), ("keep", fp), ("drop_p", C.c_float), ("params", fp),
        ("enc_off", i64 * 16), ("dec_off", i64 * 16), ("lin_off", i64 * 2), ("out", fp), ("hidden_out", fp), ("save", fp),
    ]


class Seq2SeqBwdArgs(C.Structure):
    _fields_ = [("fwd", Seq2SeqArgs), ("g_out", fp), ("dsave", fp)]


# ctypes mirror -> C struct name (checked against iplan_sizeof() of the loaded library by tests/test_abi.py)
class ObsHistArgs(C.Structure):
    _fields_ = [("K", i32), ("nA", i32), ("N", i32), ("L", i32), ("d", i32), ("obs_num", i32),
                ("obs", fp), ("agent_ids", C.c_void_p), ("slot_id", C.c_void_p), ("n_slots", C.c_void_p), ("win", fp), ("single", fp),
                ("single_s_k", i64), ("single_s_a", i64), ("err", C.c_void_p)]


STRUCT_MIRRORS = {"IplanGatSaved": GatSaved, "IplanGatFwdArgs": GatFwdArgs, "IplanGatBwdArgs": GatBwdArgs,
                  "IplanEncFwdArgs": EncFwdArgs, "IplanAcNet": AcNet, "IplanAcFeatures": AcFeatures, "IplanAcFwdArgs": AcFwdArgs,
                  "IplanAcBwdArgs": AcBwdArgs, "IplanAdamArgs": AdamArgs, "IplanWgradProblem": WgradProblem,
                  "IplanWgradArgs": WgradArgs, "IplanPpoPrepareArgs": PpoPrepareArgs, "IplanPpoLossArgs": PpoLossArgs,
                  "IplanPdecArgs": PdecArgs, "IplanBehArgs": BehArgs, "IplanMlp3Args": Mlp3Args, "IplanAdvNormArgs": AdvNormArgs, "IplanSeq2SeqArgs": Seq2SeqArgs, "IplanSeq2SeqBwdArgs": Seq2SeqBwdArgs, "IplanAcPackArgs": AcPackArgs,
                  "IplanIpcHandle": IpcHandle, "IplanP2pArgs": P2pArgs, "IplanAcXhatArgs": AcXhatArgs, "IplanAcFc1SplitArgs": AcFc1SplitArgs,
                  "IplanObsHistArgs": ObsHistArgs}
