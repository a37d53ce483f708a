"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]

def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))

for f in find("stats/**/*kernel_stats.csv"):
    print("== kernel stats (%s)" % os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-60s calls %5s  avg %12.1f ns  min %12s  max %12s  total %14s ns  %6s%%" % (
            row["Name"][:60], row["Calls"], float(row["AverageNs"]), row.get("MinNs", "-"), row.get("MaxNs", "-"),
            row["TotalDurationNs"], row["Percentage"]))

def pmc(dirname):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(dirname + "/**/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc

# per-dispatch durations of the render kernel (the first frame of a view is preceded by a short pilot launch)
for f in find("stats/**/*kernel_trace.csv"):
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_render<false" in r["Kernel_Name"]]
    if durs:
        # the timed launches are the largest group: those near the median of the upper half of all durations (below them
        # the pilot launches, above them e.g. the loop-mode launches of a mesh configuration's hardware roofline)
        upper = sorted(durs)[len(durs) // 2:]
        mid = upper[len(upper) // 2]
        full = [d for d in durs if 0.75 * mid <= d <= 1.33 * mid]
        print("== k_render<false,.> dispatches: %d, of which timed full launches %d: mean %.1f ns (min %d, max %d); shorter (pilot): %s; longer: %s" % (
            len(durs), len(full), sum(full) / len(full), min(full), max(full), [d for d in durs if d < 0.75 * mid], [d for d in durs if d > 1.33 * mid]))

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
        if "SQ_THREAD_CYCLES_VALU" in ctrs and "SQ_ACTIVE_INST_VALU" in ctrs and max(ctrs["SQ_ACTIVE_INST_VALU"]) > 0:
            # of the largest dispatch (the full render launch)
            print("    %-28s %18.3f  (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), largest dispatch)" % (
                "active lanes per VALU inst", max(ctrs["SQ_THREAD_CYCLES_VALU"]) / (64.0 * max(ctrs["SQ_ACTIVE_INST_VALU"]))))

# HBM bytes of one full render launch for bench.py's roofline.traffic: WRITE_SIZE + 2 x FETCH_SIZE of the largest
# dispatch (both counters are in KB; gfx950 tallies wide reads at half their bytes, so the doubled figure is an upper bound)
if len(sys.argv) > 2:
    import json
    fetch = max((max(v["FETCH_SIZE"]) for k, v in pmc("pmc_fetch").items() if "k_render<false" in k), default=None)
    write = max((max(v["WRITE_SIZE"]) for k, v in pmc("pmc_write").items() if "k_render<false" in k), default=None)
    if fetch is not None and write is not None:
        entry = {"bytes_per_launch": (write + 2.0 * fetch) * 1024.0, "fetch_kb": fetch, "write_kb": write,
                 "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, largest k_render<false,...> dispatch: WRITE_SIZE + 2 x FETCH_SIZE (profiles/r2, %s)" % os.path.basename(out)}
        path = os.path.join(out, "traffic_%s.json" % sys.argv[2])
        json.dump({sys.argv[2]: entry}, open(path, "w"), indent=1)
        print("== wrote %s: %.1f MB per launch" % (path, entry["bytes_per_launch"] / 1e6))
