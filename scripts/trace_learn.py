"""Kernel sequence of ONE behaviour-learn phase of a bench cycle (between a rollout's last actor/critic launch and the next
rollout's first launch) from a rocprofv3 kernel trace: python scripts/trace_learn.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((r["Kernel_Name"].split("(")[0].replace("iplan::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
              r.get("Stream_Id", ""), r.get("Queue_Id", "")) for r in rows), key=lambda r: r[1])
ge = [i for i, k in enumerate(ks) if "gat_enc" in k[0]]
# rollout boundaries: gaps > 2 ms between consecutive fused launches
bounds = [j for i, j in zip(ge, ge[1:]) if ks[j][1] - ks[i][2] > 2_000_000]
j1 = bounds[len(bounds) // 2]                     # first launch of some mid-run rollout
i0 = max(i for i in ge if i < j1)                 # last fused launch of the rollout before it
t0 = ks[i0][2]
print(f"learn phase: {(ks[j1][1] - t0) / 1e6:.2f} ms")
agg = {}
for k in ks[i0 + 1:j1]:
    if k[2] - k[1] > 150_000 or "beh_" in k[0]:
        print(f"{k[0][:46]:46s} start {(k[1] - t0) / 1e6:7.3f}  end {(k[2] - t0) / 1e6:7.3f}  dur {(k[2] - k[1]) / 1e6:6.3f} ms  q{k[4]}"
              + (("   " + k[0][:220]) if k[0].startswith("at::") else ""))
    a = agg.setdefault(k[0][:46], [0, 0])
    a[0] += 1; a[1] += k[2] - k[1]
print("-- totals in the phase")
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:16]:
    print(f"{n:46s} x{c:4d}  {d / 1e6:7.3f} ms")
