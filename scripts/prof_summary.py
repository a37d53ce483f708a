"""Scratch: prints the render kernel's durations and PMC sums (last dispatch) of scripts/prof_script.sh outputs."""
import csv, glob, collections, sys
for tag in sys.argv[1:]:
    print("==", tag)
    f = glob.glob("gpurun_out/prof_%s/stats/*/*_kernel_trace.csv" % tag)[0]
    for r in csv.DictReader(open(f)):
        if 'k_render' in r['Kernel_Name']:
            print(r['Kernel_Name'][-50:], 'grid', r['Grid_Size_X'], 'wg', r['Workgroup_Size_X'], 'lds', r.get('LDS_Block_Size'), 'vgpr', r.get('VGPR_Count'),
                  'ms %.3f' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
    for sub in ('pmc', 'pmc2'):
        fs = glob.glob("gpurun_out/prof_%s/%s/*/*_counter_collection.csv" % (tag, sub))
        if not fs: continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(fs[0])):
            if 'k_render' in r['Kernel_Name']:
                agg[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        last = sorted(agg, key=int)[-1]
        print(sub, {k: "%.4g" % v for k, v in sorted(agg[last].items())})
