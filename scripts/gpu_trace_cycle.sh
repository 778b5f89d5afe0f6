#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=gpurun_out/trace; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/p" -o cyc -- python "$R/bench.py" --in-process --steps 3 --warmup 1 --no-cpu-baseline > "$R/$O/bench.json" 2> "$R/$O/bench.err" < /dev/null )
f=$(find $O/p -name "*kernel_trace.csv" | head -1)
python scripts/trace_busy.py $f | tee $O/busy.txt
python - "$f" <<'PY' | tee $O/seq.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((r["Kernel_Name"].split("(")[0].replace("iplan::",""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", ""), r.get("Queue_Id","")) for r in rows), key=lambda r: r[1])
gat = [i for i, k in enumerate(ks) if "gat_enc_fwd" in k[0]]
i0 = gat[len(gat) // 2 + 40]            # mid-rollout somewhere
t0 = ks[i0][1]
for k in ks[i0:i0 + 16]:
    print(f"{k[0][:44]:44s} start {(k[1] - t0) / 1e3:9.1f} us  end {(k[2] - t0) / 1e3:9.1f} us  dur {(k[2] - k[1]) / 1e3:8.1f}  stream {k[3]} queue {k[4]}")
PY
rm -rf $O/p
