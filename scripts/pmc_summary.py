"""Average PMC counter values per kernel from rocprofv3 --pmc csv outputs (one sub-directory per pass).

    python scripts/pmc_summary.py <dir with p1/ p2/ ...>                       -> text table on stdout
    python scripts/pmc_summary.py <dir> --json out.json                        -> also {kernel: {counter: [avg per dispatch, dispatches]}}
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("iplan::") and "iplan" not in k:
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        s, n = agg[k][c]
        print(f"    {c:32s} avg/dispatch {s / n:16.1f}   (n={n})")
if "--json" in sys.argv:
    out = {k: {c: [v[0] / v[1], v[1]] for c, v in cs.items()} for k, cs in agg.items()}
    with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
