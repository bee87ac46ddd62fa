"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter per dispatch."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "sf::" not in r["Kernel_Name"]: continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "n=%d" % len(next(iter(cs.values()))))
