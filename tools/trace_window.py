"""Launch timeline of the kernels in front of the first launch of a given kernel in a rocprofv3 kernel trace:
usage: trace_window.py r_kernel_trace.csv <kernel-name-substring | END> <k_results_init launches to go back>."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "k_" not in n:
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "k_" + n.split("k_")[1].split("(")[0], r["Queue_Id"], r["Stream_Id"]))
ev.sort()
stop = len(ev) if sys.argv[2] == "END" else next(i for i, e in enumerate(ev) if sys.argv[2] in e[2])
inits = [i for i, e in enumerate(ev[:stop]) if e[2] == "k_results_init"]
s = inits[-int(sys.argv[3])]
while ev[s][2] != "k_meta":
    s -= 1
t0 = ev[s][0]
for e in ev[s:stop]:
    print("%8.2f %8.2f %7.2f %-28s q%s s%s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2], e[3], e[4]))
