"""Summarise a rocprofv3 rocpd database (the default output of --pmc on ROCm 7.2): per kernel, mean counter value and
mean duration per dispatch.   python scripts/pmc_db_summary.py gpurun_out/pmc_x/**/*.db"""
import collections, glob, sqlite3, sys
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        cur = sqlite3.connect(f).cursor()
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for name, cnt, val, dur in cur.execute("select name, counter_name, counter_value, duration from pmc_events"):
            if "sf::" not in name: continue
            k = name.split("(")[0].replace("void ", "")[:60]
            acc[k][cnt].append(val); acc[k]["duration_us"].append(dur / 1e3)
        for k, cs in acc.items():
            n = len(next(iter(cs.values())))
            print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, "n=%d" % n)
