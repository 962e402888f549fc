"""Launch timeline of the last full step in a rocprofv3 kernel trace of bench.py (milliseconds from the step's first kernel):
usage: trace_last_step.py r_kernel_trace.csv [parts per step = 4] [single-part runs after the steps = 2]."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # bench.py's roofline.alone leg: one part, twice
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "k_" not in n:
        continue
    short = n.split("k_")[1].split("(")[0].split("<")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "k_" + short, r["Queue_Id"], r["Stream_Id"]))
ev.sort()
idx = [i for i, e in enumerate(ev) if e[2] == "k_results_init"]


def first_meta_before(i):
    while ev[i][2] != "k_meta":
        i -= 1
    return i


s = first_meta_before(idx[-(nparts + extra)])
e_ = first_meta_before(idx[-extra]) if extra else len(ev)
t0 = ev[s][0]
for e in ev[s:e_]:
    print("%8.2f %8.2f %7.2f %-18s q%s s%s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2], e[3], e[4]))
