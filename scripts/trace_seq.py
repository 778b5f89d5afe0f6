"""Print the kernel sequence (start offset, duration, stream) following the LAST launch of a given kernel in a
rocprofv3 kernel trace:  python scripts/trace_seq.py <kernel_trace.csv> <kernel substring> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", "")) for r in rows),
            key=lambda r: r[1])
idx = [i for i, k in enumerate(ks) if sys.argv[2] in k[0]]
i0 = idx[-1]
t0 = ks[i0][1]
for k in ks[i0:i0 + int(sys.argv[3]) if len(sys.argv) > 3 else i0 + 14]:
    print(f"{k[0][:60]:60s} start {(k[1] - t0) / 1e6:8.3f} ms  dur {(k[2] - k[1]) / 1e6:8.3f} ms  stream {k[3]}")
