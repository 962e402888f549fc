"""Kernels and memory copies of a rocprofv3 run (--kernel-trace --memory-copy-trace) on one time axis."""
import csv, sys, glob, re
d = sys.argv[1]
t_from = float(sys.argv[2]) if len(sys.argv) > 2 else None
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:24], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        b = int(r.get("Size", r.get("Bytes", 0)) or 0)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "%s %.1f MB" % (r.get("Direction", "copy"), b / 1e6), ""))
ev.sort()
t0 = ev[0][0]
for s, e, name, q in ev:
    ts = (s - t0) / 1e6
    if t_from is not None and ts < t_from:
        continue
    if (e - s) / 1e6 < 0.3:
        continue
    print("%9.2f %9.2f %8.2f  %s %s" % (ts, (e - t0) / 1e6, (e - s) / 1e6, name, q))
