"""Summarise rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch (optionally / div)."""
import csv, sys, collections, glob
div = 1.0
args = []
for a in sys.argv[1:]:
    if a.startswith("--div="): div = float(a[6:])
    else: args.append(a)
for path in args:
    for f in (glob.glob(path + "/*/*counter_collection.csv") + glob.glob(path + "/*counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "tde::" not in k or "reset" in k: continue
            print(path, k)
            for c, v in sorted(d.items()):
                print(f"   {c:24s} n={len(v):3d} mean/div={sum(v)/len(v)/div:14.1f}")
