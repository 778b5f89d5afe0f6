"""Host-side launch wrappers: torch tensors in, C-ABI structs out.

Each function only (a) allocates outputs with torch, (b) fills the argument struct with raw device
pointers / strides, (c) calls the entry point on torch's current stream.  No arithmetic happens
here.  ``lib`` defaults to the gfx950 build; the CPU test-suite passes the host-emulated build of
the same kernel sources instead.
"""
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


def gat_forward(arena, src0, src1, h_prev, noise, tau=0.01, save=False, out=None, lib=None):
    """GAT_Net.forward for all nets.  src0 [n_nets,B,N,d0], src1 [n_nets,B,N,d1] or None,
    h_prev [n_nets,B,N,A] (first two dims may be arbitrarily strided views), noise
    [n_nets,B,N,N-1,2] contiguous.  Returns (out [n_nets,B,N,A], saved dict or None)."""
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
    saved = None
    if save:
        H = A
        saved = dict(
            h_enc=torch.empty(n_nets, B, N, H, device=dev),
            gru=torch.empty(n_nets, B, 2, N, N - 1, 5, H, device=dev),
            qkv=torch.empty(n_nets, B, 3, N, A, device=dev),
            soft=torch.empty(n_nets, B, N, N - 1, device=dev),
            hard=torch.empty(n_nets, B, N, N - 1, device=dev),
            x=torch.empty(n_nets, B, N, A, device=dev),
            cell=torch.empty(n_nets, B, N, 4, A, device=dev),
        )
        for k, v in saved.items():
            setattr(a.saved, k, v.data_ptr())
    lib.call("iplan_gat_fwd", a, L.current_stream(dev))
    return out, saved
