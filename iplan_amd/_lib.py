"""ctypes binding of the C ABI declared in include/iplan_hip.h.

``get_lib()`` loads the gfx950 build (``iplan_amd/libiplan_hip.so``) and raises if it is missing:
there is no CPU or PyTorch fallback anywhere in the product path.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libiplan_hip.so")

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
        ("off", i64 * GAT_NPARAM), ("tau", C.c_float), ("saved", GatSaved),
    ]


class IplanError(RuntimeError):
    pass


class Lib:
    """Thin typed wrapper around a loaded libiplan_*.so."""

    def __init__(self, cdll):
        self.c = cdll
        cdll.iplan_last_error.restype = C.c_char_p
        cdll.iplan_version.restype = C.c_int
        for name in ("iplan_gat_fwd",):
            fn = getattr(cdll, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p]

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
