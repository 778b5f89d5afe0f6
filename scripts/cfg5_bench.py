"""BASELINE.json config 5: "Synthetic 64 agents x 63 neighbours x obs_dim 128, GAT+behavior forward/backward only, rocprof
roofline run" -- command-line front of iplan_amd/config5.py (the measurement itself; bench.py appends the same rows to its
JSON line as ``config5``).

    python scripts/cfg5_bench.py [--B 32 256] [--pieces gat beh] [--json out.json]

Run it under rocprofv3 --kernel-trace --stats (and the separate --pmc passes of scripts/gpu_pmc.sh) for the per-kernel numbers
committed under profiles/.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iplan_amd.config5 import measure  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, nargs="+", default=[32, 256])
    ap.add_argument("--pieces", nargs="+", default=["gat", "beh"])
    ap.add_argument("--json", default=None)
    opt = ap.parse_args()
    rows = measure(opt.B, opt.pieces)
    for r in rows:
        print(f"B={r['B']:4d} {r['piece']:32s} {r['ms']:9.3f} ms  {r['gflop']:10.1f} GFLOP  {r['tflops']:7.2f} TFLOP/s  {100 * r['frac']:5.1f} % of fp32-MFMA peak")
    if opt.json:
        with open(opt.json, "w") as fjs:
            json.dump(rows, fjs, indent=1)


if __name__ == "__main__":
    main()
