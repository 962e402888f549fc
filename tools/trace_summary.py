"""Summary of a rocprofv3 rocpd database (kernel + memory-copy trace): for a time window, how long the GPU
had at least one kernel running, per-kernel totals, and the copies of each direction.
usage: python tools/trace_summary.py results.db [t0_ms t1_ms]   (times relative to the first kernel)"""
import re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
K = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
M = list(cur.execute("select name, start, end, size from memory_copies order by start"))
base = K[0][1]
lo = float(sys.argv[2]) * 1e6 + base if len(sys.argv) > 2 else base
hi = float(sys.argv[3]) * 1e6 + base if len(sys.argv) > 3 else K[-1][2]


def short(n):
    m = re.search(r"(k_\w+)(<[^>]*>)?", n)
    return m.group(0) if m else n[:32]


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs; cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


kk = [(n, max(s, lo), min(e, hi), st) for n, s, e, st in K if e > lo and s < hi]
print("window %.1f ms; some kernel running %.1f ms" % ((hi - lo) / 1e6, union([(s, e) for _, s, e, _ in kk]) / 1e6))
agg = {}
for n, s, e, st in kk:
    a = agg.setdefault(short(n), [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
for k, a in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print("  %-32s calls %5d  total %9.2f ms  avg %8.3f" % (k, a[0], a[1], a[1] / a[0]))
for d in ("HOST_TO_DEVICE", "DEVICE_TO_HOST"):
    mm = [(max(s, lo), min(e, hi), sz) for n, s, e, sz in M if d in n and e > lo and s < hi]
    print("  %s: %d copies, %.2f GB, engine busy %.1f ms" % (d, len(mm), sum(x[2] for x in mm) / 1e9, union([(s, e) for s, e, _ in mm]) / 1e6))
