#!/bin/bash
# Vector-memory pipeline counters (TA / TCP, one pass each) of the render kernels of any scratch script.
# Usage: scripts/pmc_tcp.sh <tag> <script.py> [args]   -> gpurun_out/prof_<tag>/tcp.txt
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta -- python3 "$@" > $OUT/pmc_ta.log 2>&1 || { echo "ta failed"; tail -5 $OUT/pmc_ta.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $OUT/pmc_tcp1 -- python3 "$@" > $OUT/pmc_tcp1.log 2>&1 || { echo "tcp1 failed"; tail -5 $OUT/pmc_tcp1.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq -- python3 "$@" > $OUT/pmc_sq.log 2>&1 || { echo "sq failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
python3 - $OUT > $OUT/tcp.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for d in ("pmc_ta", "pmc_tcp1", "pmc_sq"):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, d, "**/*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    for k, ctrs in acc.items():
        if "k_render" not in k: continue
        print(k[:90])
        for c, vals in sorted(ctrs.items()):
            print("  %-10s %-36s %s" % (d, c, " ".join("%.4g" % v for _, v in sorted(vals))))
PY
cat $OUT/tcp.txt
