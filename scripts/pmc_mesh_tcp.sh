#!/bin/bash
# Vector-memory pipeline counters of the mesh render kernel (separate passes: 2 TA / 4 TCP counters each).
set -u
OUT=gpurun_out/prof_${1:-tcp}
SPP=${2:-16}
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 scripts/bvh_check.py 1920 $SPP"
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta -- $B > $OUT/pmc_ta.log 2>&1 || { echo "ta failed"; tail -5 $OUT/pmc_ta.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc_tcp1 -- $B > $OUT/pmc_tcp1.log 2>&1 || { echo "tcp1 failed"; tail -5 $OUT/pmc_tcp1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp2 -- $B > $OUT/pmc_tcp2.log 2>&1 || { echo "tcp2 failed"; tail -5 $OUT/pmc_tcp2.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for d in ("pmc_ta", "pmc_tcp1", "pmc_tcp2"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, d, "**/*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, ctrs in acc.items():
        if "k_render" not in k: continue
        for c, vals in sorted(ctrs.items()):
            print("%-14s %-40s %s" % (d, c, " ".join("%.4g" % v for v in vals)))
PY
