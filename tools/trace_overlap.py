"""Print start/end (ms, relative) of kernels in a rocprofv3 kernel_trace.csv (last N rows)."""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s q=%s start %9.3f end %9.3f dur %8.3f grid %s lds %s vgpr %s" % (
        (re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]) or [r["Kernel_Name"][:28]])[0], r.get("Queue_Id", "?"), (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6,
        r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?")))
