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


# ------------------------------------------------------------------------------------------------ hardware queues
# HIP runs a process's streams on GPU_MAX_HW_QUEUES hardware queues (4 by default), assigned in creation order; two streams on one
# queue execute in submission order -- they do not overlap, whatever the events between them say.  The training cycle keeps about
# eight streams (main, two learner streams, the encoder's forward / BPTT side streams, the deferred decoder update's, ...), so some
# MUST share, and WHICH ones do decides the cycle: with the encoder-BPTT stream on the main stream's queue the encoder ranges run
# behind the decoder ranges instead of beside them (+20 ms per cycle).  Until round 6 the good assignment was an accident of creation
# order -- a live RCCL communicator (every N > 1 rank) creates its streams first and shifts it: 264 -> 271 ... 283 ms by queue count
# (profiles/r06_notes.md section 8, kernel traces with the queue of every kernel).  The streams that must overlap the main stream are
# therefore PROBED: ``distinct_stream`` hands out a pool stream that demonstrably does not share a queue with the given ones.
_probe_state = {}


def shares_queue(a, b, device):
    """Do streams ``a`` and ``b`` execute on the same hardware queue?  A ~0.5 ms spin on ``a``, then a one-element op on ``b``: on one
    queue the op cannot finish before the spin does.  Synchronises the device (start-up only)."""
    device = torch.device(device)
    st = _probe_state.get(str(device))
    if st is None:
        st = _probe_state[str(device)] = dict(flag=torch.zeros(1, device=device), cycles=None)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):                                   # (first call: module load)
            e0.record()
            torch.cuda._sleep(1_000_000)
            e1.record()
            torch.cuda.synchronize(device)
        ms = max(e0.elapsed_time(e1), 1e-3)
        st["cycles"] = int(min(max(1_000_000 * 0.5 / ms, 100_000), 50_000_000))     # ~0.5 ms
    torch.cuda.synchronize(device)
    a0, a1, b1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(a):
        a0.record(a)
        torch.cuda._sleep(st["cycles"])
        a1.record(a)
    with torch.cuda.stream(b):
        st["flag"].add_(1.0)
        b1.record(b)
    torch.cuda.synchronize(device)
    return a0.elapsed_time(b1) > 0.5 * a0.elapsed_time(a1)


def probe_mode():
    """IPLAN_QUEUE_PROBE =
      verify (default): streams are created in the old order, and AFTER the first training cycle (every stream has had its first use,
              i.e. its queue) the pairs that must not share -- main vs the encoder's two side streams vs the prediction learner's -- are
              probed (harness.SyntheticLoop._verify_queues); a stream that does share is replaced (``distinct_stream``).  Nothing is
              touched where the creation order came out right (a process without RCCL: traced main | prediction | encoder BPTT | encoder
              forward + PPO, 19 runs in 20), so the single-GPU line times what every evidence series timed; with a live RCCL group
              (every N > 1 rank) the encoder-BPTT and prediction streams get replaced: 267 ms against 276 unprobed
      full:   every stream that overlaps another is probed against it AT CREATION, roles created up front (harness.cycle), the PPO update on
              the prediction learner's stream: 264-270 ms whatever RCCL or GPU_MAX_HW_QUEUES did (unprobed: 262 ... 289 ms); 0.8 % slower
              than a good creation-order assignment (the probes' launches change the order of first use)
      min:    creation order, the side streams checked against the main stream's queue only, at creation
      0:      creation order, unchecked"""
    m = os.environ.get("IPLAN_QUEUE_PROBE", "verify")
    return m if m in ("min", "full", "verify") else "0"


def _probing():
    return probe_mode() in ("min", "full")


def distinct_stream(device, avoid=(), tries=12, priority=0, force=False):
    """A torch pool stream on ``device`` that shares a hardware queue with none of the streams in ``avoid`` (best effort: after
    ``tries`` candidates the last one is returned).  CPU devices / IPLAN_QUEUE_PROBE=0: a plain pool stream."""
    device = torch.device(device)
    if device.type != "cuda" or not (force or _probing()) or not hasattr(torch.cuda, "_sleep"):
        return torch.cuda.Stream(device, priority=priority)
    avoid = [s for s in avoid if s is not None]
    held = []                                                # (candidates stay referenced until the choice is made: distinct pool entries)
    for _ in range(max(1, tries)):
        c = torch.cuda.Stream(device, priority=priority)
        held.append(c)
        try:
            if not any(shares_queue(o, c, device) for o in avoid):
                return c
        except Exception:                                    # noqa: BLE001 -- a probe problem must never cost the run: plain stream
            return c
    return held[-1]


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
