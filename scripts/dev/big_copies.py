"""Kernel-trace helper: the long __amd_rocclr_copyBuffer launches of a rocprofv3 kernel trace with their neighbours in time.
    python scripts/dev/big_copies.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((r["Kernel_Name"].split("(")[0].replace("iplan::","").replace("void ",""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id",""), r.get("Grid_Size_X", r.get("Grid_Size","")), r.get("Workgroup_Size_X","")) for r in rows), key=lambda r: r[1])
big = [i for i,k in enumerate(ks) if "copyBuffer" in k[0] and k[2]-k[1] > 300_000]
print("copies > 0.3 ms:", len(big))
for i in big[-6:]:
    print("---- copy dur %.3f ms  grid %s wg %s q%s" % ((ks[i][2]-ks[i][1])/1e6, ks[i][4], ks[i][5], ks[i][3]))
    for k in ks[max(0,i-6):i+7]:
        print("   %-44s start %+9.3f ms dur %7.3f ms q%s" % (k[0][:44], (k[1]-ks[i][1])/1e6, (k[2]-k[1])/1e6, k[3]))
