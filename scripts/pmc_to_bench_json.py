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
merged = {}
for f in files:
    for k, cs in json.load(open(f)).items():
        merged.setdefault(k, {}).update(cs)


def find(sub):
    ks = [k for k in merged if sub in k]
    assert ks, (sub, sorted(merged))
    return ks


def bytes_of(k):
    cs = merged[k]
    return (2.0 * cs["FETCH_SIZE"][0] + cs["WRITE_SIZE"][0]) * 1024.0, cs["FETCH_SIZE"][1]


per_launch = {}
for key, sub in (("gat_enc_fwd_kernel", "gat_enc_fwd_kernel"), ("gat_enc_ac_fwd_kernel", "gat_enc_ac_fwd_kernel"),
                 ("beh_dec_bwd_kernel", "beh_dec_bwd"), ("beh_dec_fwd_kernel", "beh_dec_fwd"),
                 ("beh_enc_bwd_kernel", "beh_enc_bwd_kernel"), ("ac_fwd_kernel:train", "ac_fwd_kernel<2, true"),
                 ("ac_fc1_split_fwd", "ac_fc1_split_fwd_kernel"), ("ac_fc1_split_wgrad", "ac_fc1_split_wgrad_kernel"),
                 ("ac_bwd_tail_kernel", "ac_bwd_tail_kernel")):
    if key.startswith("gat_enc") and not [k for k in merged if sub in k]:
        continue                                     # (the fused three-part launch replaces most gat_enc_fwd launches, or is switched off)
    k = find(sub)[0]
    b, n = bytes_of(k)
    per_launch[key] = dict(bytes=int(b), dispatches=n, kernel=k)
# iplan_wgrad over one whole decoder BPTT: every wgrad kernel of the behaviour piece, per learn() (one beh_enc_grad launch each).
# The piece runs learn(defer_decoder=True) (scripts/gpu_pmc_all.sh: MB_DEFER=1), so these ARE the kernels of the ONE deferred
# iplan_wgrad call that bench.py times as "iplan_wgrad:beh_dec" (the encoder accumulates its weight gradients in its BPTT kernel
# and launches no wgrad kernel): traffic / us_per_launch of that line is a bandwidth that can be compared with the HBM peak.
learns = merged[find("beh_enc_grad_kernel")[0]]["FETCH_SIZE"][1]
tot = 0.0
for k in find("wgrad_"):
    b, n = bytes_of(k)
    tot += b * n
per_launch["iplan_wgrad:beh_dec"] = dict(bytes=int(tot / learns), dispatches=learns, kernel="wgrad_partial_*kernel<*> + wgrad_reduce_kernel of the deferred decoder update, per learn()")
out = dict(series=series, csrc_sha16=csrc_sha16(), envs_per_gpu=envs, per_launch=per_launch, counters=merged)
path = os.path.join(ROOT, "profiles", f"{series}_pmc_summary.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
print(path, {k: v["bytes"] for k, v in per_launch.items()})
