"""Tie bench.py's ``cpu_baseline`` (the oracle, kind "port") to the REAL reference: time both on identical inputs with the same
sampling protocol (oracle/cpu_baseline.py) in the build container, where /root/reference exists, and write the raw numbers and
the ratio to profiles/r06_cpu_baseline_anchor.json (behaviour leg = all agents at the bench's own width Eb = 32, PPO leg = one agent x
all 22 950 rows; re-run whenever oracle/ changes -- round 3's file is kept for the history).      python scripts/anchor_cpu_baseline.py [--envs 32]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.cpu_baseline import measure  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=32)
ap.add_argument("--eb", type=int, default=32)
ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_cpu_baseline_anchor.json"))
opt = ap.parse_args()
cores = min(os.cpu_count() or 1, 16)
res = {b: measure(b, opt.envs, cores, Eb=opt.eb) for b in ("oracle", "reference")}
out = dict(host=f"{os.cpu_count()} logical cores (build container), {cores} torch threads", envs=opt.envs,
           oracle=res["oracle"], reference=res["reference"],
           reference_over_oracle_time={k: res["reference"]["seconds"][k] / res["oracle"]["seconds"][k] for k in res["oracle"]["seconds"]},
           oracle_over_reference_env_steps_per_s=res["oracle"]["value"] / res["reference"]["value"])
with open(opt.out, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: out[k] for k in ("reference_over_oracle_time", "oracle_over_reference_env_steps_per_s")}, indent=1))
print("oracle", res["oracle"]["value"], "reference", res["reference"]["value"], "env-steps/s")
