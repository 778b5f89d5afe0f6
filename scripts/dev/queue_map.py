"""Which hardware queue did each kernel family of a bench cycle run on (rocprofv3 kernel trace, column Queue_Id)?  The cycle's streams
share HIP's GPU_MAX_HW_QUEUES hardware queues; which of them share one decides what serialises (profiles/r06_notes.md).
    python scripts/dev/queue_map.py <kernel_trace.csv>"""
import csv
import sys
from collections import Counter, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
fam = defaultdict(Counter)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("iplan::", "").replace("void ", "")
    n = n.split("<")[0]
    fam[n][r.get("Queue_Id", "?")] += 1
tot = Counter()
for n, c in fam.items():
    for q, k in c.items():
        tot[q] += k
print("queues:", dict(tot))
for n, c in sorted(fam.items(), key=lambda kv: -sum(kv[1].values()))[:40]:
    print(f"{n[:44]:44s} " + "  ".join(f"q{q}:{k}" for q, k in sorted(c.items())))
