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

# Derived utilisation of the TIMED launches of the configuration's render kernel (and of the largest dispatch of every other
# render kernel).  SIMDs = 256 CUs x 4; a wave64 VALU instruction occupies its SIMD's issue port for 2 cycles on gfx950 (32
# lanes per clock), so
#   VALU issue %  = 2 x SQ_INSTS_VALU       / (SIMDs x cycles)   instructions issued against the 2-cycle ceiling
#   VALUBusy %    = 2 x SQ_ACTIVE_INST_VALU / (SIMDs x cycles)   the same from the busy counter (includes multi-pass instructions)
#   VALU lanes %  = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)   active lanes per issued instruction
#   wait %        = SQ_WAIT_ANY / SQ_WAVE_CYCLES                  share of resident wave-cycles spent waiting
# cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs).  Every pass runs the same sequence of dispatches, so the timed
# launches are picked in the sq2 pass (cycles within 0.6 ... 1.4 of the traced duration of the timed full launches) and the same
# positions are read in the other passes.
def pmc_ordered(dirname):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(dirname + "/**/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    return {k: {c: [v for _, v in sorted(vals)] for c, vals in ctrs.items()} for k, ctrs in acc.items()}


full_ns = None
for f in find("stats/**/*kernel_trace.csv"):
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_render<false" in r["Kernel_Name"]]
    if durs:
        upper = sorted(durs)[len(durs) // 2:]
        mid = upper[len(upper) // 2]
        sel = [d for d in durs if 0.75 * mid <= d <= 1.33 * mid]
        full_ns = sum(sel) / len(sel)
sq, sq2, fe, wr = pmc_ordered("pmc_sq"), pmc_ordered("pmc_sq2"), pmc_ordered("pmc_fetch"), pmc_ordered("pmc_write")
SIMDS, MEASURED_RATE = 1024.0, 0.87e9  # measured plain-VALU issue rate per SIMD under load (scripts/ubench/valu_rate.hip), instructions/s


def timed_positions(k):
    """positions (in dispatch order) of the timed full launches among the dispatches of kernel k"""
    g = sq2.get(k, {}).get("GRBM_GUI_ACTIVE", [])
    if not g:
        return []
    pos = [i for i, v in enumerate(g) if full_ns and 0.6 <= (v / 8.0) / (full_ns * 2.4) <= 1.4] if "k_render<false" in k else []
    return pos or [max(range(len(g)), key=lambda i: g[i])]


def mean_at(vals, pos):
    vals = [vals[i] for i in pos if i < len(vals)]
    return sum(vals) / len(vals) if vals else float("nan")


rows = []
for k, c in sq.items():
    if "k_render" not in k or k not in sq2 or "SQ_INSTS_VALU" not in c:
        continue
    pos = timed_positions(k)
    if not pos:
        continue
    cycles = mean_at(sq2[k]["GRBM_GUI_ACTIVE"], pos) / 8.0
    insts, busy = mean_at(c["SQ_INSTS_VALU"], pos), mean_at(c.get("SQ_ACTIVE_INST_VALU", []), pos)
    lanes = mean_at(c["SQ_THREAD_CYCLES_VALU"], pos) / (64.0 * busy) if busy == busy and busy else float("nan")
    wait = mean_at(c["SQ_WAIT_ANY"], pos) / mean_at(c["SQ_WAVE_CYCLES"], pos) if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c else float("nan")
    rows.append((k, cycles, 200.0 * insts / (SIMDS * cycles), 200.0 * busy / (SIMDS * cycles), 100.0 * lanes, 100.0 * wait,
                 100.0 * insts / (SIMDS * MEASURED_RATE * cycles / 2.4e9), len(pos), insts))
util = {}
timed_kernel = None
if rows:
    print("== derived utilisation (render kernels: mean over the timed full launches; others: largest dispatch)")
    for k, cycles, issue, busy, lanes, wait, vs_measured, n, insts in rows:
        print("  %s" % k[:100])
        print("    dispatches %d  cycles %.4g (%.3f ms at 2.4 GHz)  SQ_INSTS_VALU %.4g  VALU issue %.1f %%  VALUBusy %.1f %%  active lanes %.1f %%  waiting %.1f %% of wave-cycles  "
              "(issue rate = %.0f %% of the measured 0.87 G inst/s/SIMD plain-VALU rate)" % (n, cycles, cycles / 2.4e6, insts, issue, busy, lanes, wait, vs_measured))
    # the timed render kernel of the configuration: the k_render<false,...> build with the most cycles in its timed launches
    best = max((r for r in rows if "k_render<false" in r[0]), key=lambda r: r[1] * r[7], default=None)
    if best is not None:
        timed_kernel = best[0]
        util = {"kernel": best[0].split("::")[-1].replace("(RenderParams)", ""), "valu_issue_pct": round(best[2], 1), "valu_busy_pct": round(best[3], 1),
                "active_lanes_pct": round(best[4], 1), "waiting_pct_of_wave_cycles": round(best[5], 1), "valu_wave_instructions": best[8]}

# HBM bytes of one timed render launch for bench.py's roofline.traffic: WRITE_SIZE + 2 x FETCH_SIZE (both counters are in KB;
# gfx950 tallies wide reads at half their bytes, so the doubled figure is an upper bound)
if len(sys.argv) > 2 and timed_kernel is not None:
    import json
    pos = timed_positions(timed_kernel)
    fetch = mean_at(fe.get(timed_kernel, {}).get("FETCH_SIZE", []), pos)
    write = mean_at(wr.get(timed_kernel, {}).get("WRITE_SIZE", []), pos)
    if fetch == fetch and write == write:
        entry = {"bytes_per_launch": (write + 2.0 * fetch) * 1024.0, "fetch_kb": fetch, "write_kb": write, "utilisation": util,
                 "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, mean over the timed k_render<false,...> launches: WRITE_SIZE + 2 x FETCH_SIZE (profiles/r3, %s)" % os.path.basename(out)}
        path = os.path.join(out, "traffic_%s.json" % sys.argv[2])
        json.dump({sys.argv[2]: entry}, open(path, "w"), indent=1)
        print("== wrote %s: %.1f MB per launch" % (path, entry["bytes_per_launch"] / 1e6))
