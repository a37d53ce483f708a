"""Prints rocprofv3 counter_collection.csv as one line per dispatch (kernel, counters...)."""
import csv, glob, os, sys
from collections import defaultdict
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**/*counter_collection.csv"), recursive=True)):
    rows = defaultdict(dict)
    names = {}
    for row in csv.DictReader(open(f)):
        rows[int(row["Dispatch_Id"])][row["Counter_Name"]] = float(row["Counter_Value"])
        names[int(row["Dispatch_Id"])] = row["Kernel_Name"]
    for d in sorted(rows):
        if len(sys.argv) > 2 and sys.argv[2] not in names[d]:
            continue
        print("%4d %-40s %s" % (d, names[d].replace("void (anonymous namespace)::", "")[:40],
                                "  ".join("%s=%.4g" % (k.replace("SQ_", ""), v) for k, v in sorted(rows[d].items()))))
