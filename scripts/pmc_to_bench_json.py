"""Build profiles/<series>_pmc_summary.json -- the ONLY source of bench.py's ``roofline.traffic`` -- from the per-piece PMC
summaries of ONE build (scripts/gpu_pmc_all.sh -> scripts/pmc_summary.py --json).  The file carries the hash of the kernel
sources it was measured on (``csrc_sha16``, the same function bench.py evaluates at run time): bench.py reports traffic only
when the hashes agree, so a number can never outlive the build it came from.

    python scripts/pmc_to_bench_json.py <series> <pmc_rollout.json> <pmc_behaviour.json> <pmc_ppo.json> [envs_per_gpu]

HBM bytes of one dispatch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (on gfx950 rocprofv3's FETCH_SIZE counts 128-byte requests as
64 bytes: /opt/skills/guides/MI355X_MICROARCH.md, HBM section; separate --pmc passes for the two counters)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16  # noqa: E402

series, files = sys.argv[1], sys.argv[2:5]
envs = int(sys.argv[5]) if len(sys.argv) > 5 else 32
# The three pieces stay SEPARATE: the same kernel name occurs in more than one of them with different shapes (the wgrad kernels of
# the behaviour piece's deferred decoder update and of the PPO piece's epochs) -- round 4 merged them by name, the PPO piece's
# counters overwrote the behaviour piece's and the "iplan_wgrad:beh_dec" row came out at 31 GB / 10.8 TB/s.
PIECES = ("rollout", "behaviour", "ppo")
pieces = {p: json.load(open(f)) for p, f in zip(PIECES, files)}
merged = {f"{p}:{k}": cs for p in PIECES for k, cs in pieces[p].items()}      # (kept in the summary file, keyed by (piece, kernel))


def find(sub, piece):
    ks = [k for k in pieces[piece] if sub in k]
    assert ks, (sub, piece, sorted(pieces[piece]))
    return ks


def bytes_of(k, piece):
    cs = pieces[piece][k]
    assert cs["FETCH_SIZE"][1] == cs["WRITE_SIZE"][1], (piece, k, cs)        # both passes saw the same number of dispatches
    return (2.0 * cs["FETCH_SIZE"][0] + cs["WRITE_SIZE"][0]) * 1024.0, cs["FETCH_SIZE"][1]


per_launch = {}
for key, sub, piece in (("gat_enc_fwd_kernel", "gat_enc_fwd_kernel", "rollout"), ("gat_enc_ac_fwd_kernel", "gat_enc_ac_fwd_kernel", "rollout"),
                        ("beh_dec_bwd_kernel", "beh_dec_bwd", "behaviour"), ("beh_dec_fwd_kernel", "beh_dec_fwd", "behaviour"),
                        ("beh_enc_bwd_kernel", "beh_enc_bwd_kernel", "behaviour"), ("beh_enc_fwd_kernel", "beh_enc_fwd_kernel", "behaviour"),
                        ("ac_fwd_kernel:train", "ac_fwd_kernel<2, true", "ppo"),
                        ("ac_fc1_split_fwd", "ac_fc1_split_fwd_kernel", "ppo"), ("ac_fc1_split_wgrad", "ac_fc1_split_wgrad_kernel", "ppo"),
                        ("ac_bwd_tail_kernel", "ac_bwd_tail_kernel", "ppo")):
    if key.startswith("gat_enc") and not [k for k in pieces[piece] if sub in k]:
        continue                                     # (the fused three-part launch replaces most gat_enc_fwd launches, or is switched off)
    k = find(sub, piece)[0]
    b, n = bytes_of(k, piece)
    per_launch[key] = dict(bytes=int(b), dispatches=n, kernel=k, piece=piece)
# iplan_wgrad over one whole decoder BPTT: every wgrad kernel of the behaviour piece, per learn() (one beh_enc_grad launch each).
# The piece runs learn(defer_decoder=True) (scripts/gpu_pmc_all.sh: MB_DEFER=1), so these ARE the kernels of the ONE deferred
# iplan_wgrad call that bench.py times as "iplan_wgrad:beh_dec" (the encoder accumulates its weight gradients in its BPTT kernel
# and launches no wgrad kernel): traffic / us_per_launch of that line is a bandwidth that can be compared with the HBM peak.
# Kernels of wgrad.hip ONLY (iplan::wgrad_partial_* / wgrad_pair_bf16_kernel / wgrad_reduce_kernel; NOT ac_fc1_split_wgrad_kernel), behaviour piece ONLY.
learns = pieces["behaviour"][find("beh_enc_grad_kernel", "behaviour")[0]]["FETCH_SIZE"][1]
tot = 0.0
wg = [k for k in pieces["behaviour"] if ("wgrad_partial" in k or "wgrad_pair" in k or "wgrad_reduce" in k)]
assert wg, sorted(pieces["behaviour"])
for k in wg:
    assert "ac_fc1" not in k, k
    b, n = bytes_of(k, "behaviour")
    tot += b * n
per_launch["iplan_wgrad:beh_dec"] = dict(bytes=int(tot / learns), dispatches=learns, piece="behaviour", kernels=sorted(wg),
                                         kernel="wgrad_pair_bf16_kernel (+ wgrad_partial_*) + wgrad_reduce_kernel of the deferred decoder update, per learn()")
out = dict(series=series, csrc_sha16=csrc_sha16(), envs_per_gpu=envs, per_launch=per_launch, counters=merged)
path = os.path.join(ROOT, "profiles", f"{series}_pmc_summary.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(path, {k: v["bytes"] for k, v in per_launch.items()})
