"""Summarise rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch."""
import csv, sys, collections, glob
for path in sys.argv[1:]:
    for f in glob.glob(path + "/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "tde::" not in k: continue
            print(f, k)
            for c, v in sorted(d.items()):
                print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
