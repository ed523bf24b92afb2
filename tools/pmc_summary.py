"""Mean of each PMC counter per kernel from a rocprofv3 --pmc run directory (counter_collection.csv)."""
import collections, csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if len(sys.argv) < 3 or sys.argv[2] in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
