"""Average PMC counter values of one kernel from a rocprofv3 --pmc ... --output-format csv run.
usage: python tools/pmc_kernel.py <dir> <kernel substring>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = {}
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(f"{k:28s} {sum(v) / len(v):16.0f}  ({len(v)} dispatches)")
