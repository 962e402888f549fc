"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, sum of each counter."""
import csv, sys, glob, collections, re
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    k = m.group(1) if m else r["Kernel_Name"][:24]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, cs in acc.items():
    print(k, {c: (v, n[(k, c)]) for c, v in cs.items()})
