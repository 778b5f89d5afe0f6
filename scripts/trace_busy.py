"""GPU busy / idle time of the LAST bench cycle in a rocprofv3 kernel trace, and a per-phase timeline: the cycle is cut at the
first fused vector-step launch of a rollout.  python scripts/trace_busy.py <kernel_trace.csv>
Prints: cycle wall time, union of kernel intervals (GPU busy), idle gaps by size class, and the phases (rollout = from a
rollout's first vector-step launch -- gat_enc_ac_fwd / gat_enc_fwd -- to the end of its last one; learn = until the next rollout's first)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows), key=lambda r: r[1])
# rollouts: runs of vector-step launches; a new rollout starts when the gap since the previous one exceeds 2 ms
gat = [k for k in ks if "gat_enc" in k[0]]
starts = [gat[0][1]]
for a, b in zip(gat, gat[1:]):
    if b[1] - a[2] > 2_000_000:
        starts.append(b[1])
print("rollouts found:", len(starts))
if len(starts) < 17:
    sys.exit(0)
t0, t1 = starts[-16], starts[-8]                              # 8 rollouts = one cycle (the one before the last)
cyc = [k for k in ks if k[1] >= t0 and k[1] < t1]
iv = sorted((k[1], k[2]) for k in cyc)
busy, cur_s, cur_e, gaps = 0, iv[0][0], iv[0][1], []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
print(f"cycle wall {wall / 1e6:.1f} ms, GPU busy (union of kernels) {busy / 1e6:.1f} ms, idle {(wall - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e6), (1e6, 1e9)):
    g = [x for x in gaps if lo <= x < hi]
    print(f"  gaps {lo / 1e3:7.0f}-{hi / 1e3:7.0f} us: {len(g):6d}, total {sum(g) / 1e6:7.2f} ms")
# phases of the cycle's rollouts
for i in range(8):
    a, b = starts[-16 + i], starts[-16 + i + 1]
    last_ac = max(k[2] for k in gat if a <= k[1] < b)
    print(f"  rollout {i}: {(last_ac - a) / 1e6:6.2f} ms, then until the next rollout {(b - last_ac) / 1e6:6.2f} ms")
