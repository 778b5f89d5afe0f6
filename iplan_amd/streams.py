"""HIP streams restricted to a subset of the CUs (hipExtStreamCreateWithCUMask), wrapped for torch.

Work that nothing on the critical path waits for (the behaviour decoder's weight-gradient contraction and optimiser step,
Behavior_policy.learn(defer_decoder=True)) is bandwidth bound and would otherwise put one long-lived 372-register wave on
every SIMD of the chip the moment it starts -- the rollout kernels that follow then queue behind it.  On a masked stream it
keeps to its CUs and the rollout's workgroups land on the others.  Mask bits interleave over the 8 XCDs (bits 0..8k-1 = k CUs
per XCD; scripts/ubench/cu_mask_probe.hip), so a prefix mask stays balanced over the XCDs' L2s and memory channels."""
import ctypes as C
import os

import torch

_hip = None


def _hip_runtime():
    """the HIP runtime instance this process already uses (torch's), not a second copy"""
    global _hip
    if _hip is None:
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        _hip = C.CDLL(path or "libamdhip64.so")
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
    return _hip


def masked_stream(device, n_cus, first=0):
    """A torch stream on ``device`` that may only use the ``n_cus`` CUs of mask bits [first, first + n_cus) (multiples of 8 keep
    it spread evenly over the XCDs); a plain side stream when the mask cannot be set up (``IPLAN_NO_CU_MASK=1``, range outside
    the 256 CUs, runtime without the extension)."""
    device = torch.device(device)
    if os.environ.get("IPLAN_NO_CU_MASK") or n_cus <= 0 or first < 0 or first + n_cus > 256 or n_cus == 256:
        return torch.cuda.Stream(device)
    try:
        words = (C.c_uint32 * 8)()
        for i in range(first, first + n_cus):
            words[i // 32] |= 1 << (i % 32)
        handle = C.c_void_p()
        with torch.cuda.device(device):
            torch.cuda.current_stream()                       # (context of this device is current)
            rc = _hip_runtime().hipExtStreamCreateWithCUMask(C.byref(handle), 8, words)
        if rc != 0 or not handle.value:
            return torch.cuda.Stream(device)
        return torch.cuda.ExternalStream(handle.value, device=device)
    except Exception:                                          # noqa: BLE001  (no extension: the plain stream is always correct)
        return torch.cuda.Stream(device)


class AsyncHost:
    """Device tensor -> pinned host copy on a dedicated copy stream, ordered behind everything enqueued so far on the CURRENT
    stream (where the tensor was produced); ``get()`` waits for that copy alone.  A plain ``tensor.cpu()`` is ordered on the
    current stream: called later it would also wait for whatever the caller has enqueued there since (the next rollout)."""
    _copy_streams = {}

    def __init__(self, t):
        self.ev = None
        if t.device.type != "cuda":
            self.host = t
            return
        dev = t.device
        cs = AsyncHost._copy_streams.get(str(dev))
        if cs is None:
            cs = AsyncHost._copy_streams[str(dev)] = torch.cuda.Stream(dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        self.host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        with torch.cuda.stream(cs):
            cs.wait_event(ready)
            self.host.copy_(t, non_blocking=True)
            self.ev = torch.cuda.Event()
            self.ev.record(cs)
        t.record_stream(cs)

    def get(self):
        if self.ev is not None:
            self.ev.synchronize()
            self.ev = None
        return self.host
