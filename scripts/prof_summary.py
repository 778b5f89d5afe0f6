"""Condense a rocprofv3 (--kernel-trace --stats) result into a small text summary for profiles/.
Accepts either the rocpd sqlite DB (bench_results.db) or a *_kernel_stats.csv."""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+)", name)
    base = m.group(1) if m else name[:60]
    for tag in ("direct_copy", "exponential", "FillFunctor", "neg_kernel", "log_kernel", "scatter", "normal_kernel",
                "MulFunctor", "softmax"):
        if tag in name and not base.startswith("iplan"):
            return f"torch::{tag} [{base.split('::')[-1]}]"
    return base


def main(src, out):
    rows = []
    if src.endswith(".db"):
        c = sqlite3.connect(src).cursor()
        for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            rows.append((short(name), calls, total, avg, pct))
    else:
        for r in csv.DictReader(open(src)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                         float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    agg = {}
    for n, calls, total, avg, pct in rows:
        a = agg.setdefault(n, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += total
        a[2] += pct
    with open(out, "w") as f:
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for n, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{n},{calls},{total:.1f},{total / calls:.2f},{pct:.2f}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
