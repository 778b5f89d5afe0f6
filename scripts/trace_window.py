"""Every kernel (also the tiny ones) of two windows of a bench cycle, from a rocprofv3 kernel trace:
  * the first HEAD ms of a mid-run learn phase (from the end of a rollout's last fused GAT + encoder launch), and
  * one PPO epoch of the last train() (from one ppo_loss_kernel to the next).
python scripts/trace_window.py <kernel_trace.csv> [HEAD ms = 3.0]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
head_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
ks = sorted(((r["Kernel_Name"].split("(")[0].replace("iplan::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
              r.get("Queue_Id", "")) for r in rows), key=lambda r: r[1])


def show(sel, t0):
    for k in sel:
        print(f"  {k[0][:70]:70s} start {(k[1] - t0) / 1e3:9.1f} us  dur {(k[2] - k[1]) / 1e3:8.1f} us  q{k[3]}")


ge = [i for i, k in enumerate(ks) if "gat_enc" in k[0]]
bounds = [(i, j) for i, j in zip(ge, ge[1:]) if ks[j][1] - ks[i][2] > 2_000_000]
if bounds:
    i0, j1 = bounds[len(bounds) // 2]
    t0 = ks[i0][2]
    print(f"== learn phase head: every kernel that starts within {head_ms} ms of the end of the rollout's last fused launch (phase {(ks[j1][1] - t0) / 1e6:.2f} ms)")
    show([k for k in ks[i0 + 1:j1] if k[1] - t0 < head_ms * 1e6], t0)
    print(f"== learn phase tail / next rollout head: kernels that start in the last 1.5 ms before the next rollout's first fused launch, and 1.5 ms after")
    t1 = ks[j1][1]
    show([k for k in ks if -1.5e6 < k[1] - t1 < 1.5e6], t1)
pl = [i for i, k in enumerate(ks) if "ppo_loss_kernel" in k[0]]
if len(pl) >= 4:
    a, b = pl[-3], pl[-2]
    print(f"== one PPO epoch (ppo_loss -> ppo_loss): {(ks[b][1] - ks[a][1]) / 1e3:.1f} us, {b - a} kernels")
    show(ks[a:b + 1], ks[a][1])
    # the whole train(): from the critic pass before the first ppo_loss of the last block to the last Adam step
    blocks = [pl[0]]
    for i, j in zip(pl, pl[1:]):
        if ks[j][1] - ks[i][1] > 20_000_000:
            blocks.append(j)
    first = blocks[-1]
    print(f"== last train(): first ppo_loss -> last ppo_loss {(ks[pl[-1]][1] - ks[first][1]) / 1e6:.2f} ms over {sum(1 for i in pl if i >= first)} epochs")
