"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]

def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))

for f in find("stats/**/*kernel_stats.csv"):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls %5s  avg %12.1f ns  total %14s ns  %6s%%" % (
            row["Name"][:60], row["Calls"], float(row["AverageNs"]), row["TotalDurationNs"], row["Percentage"]))

def pmc(dirname):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(dirname + "/**/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc

for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_tcp"):
    acc = pmc(d)
    if not acc:
        continue
    print("== %s (mean per dispatch)" % d)
    for k, ctrs in acc.items():
        if "k_render" not in k and "tonemap" not in k:
            continue
        print("  kernel %s" % k[:70])
        for c, vals in sorted(ctrs.items()):
            print("    %-28s %18.1f  (n=%d, max %.1f)" % (c, sum(vals) / len(vals), len(vals), max(vals)))
