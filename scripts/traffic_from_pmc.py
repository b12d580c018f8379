"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4
slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 — MI355X_MICROARCH.md 'rocprofv3 PMC slots').  Units: KiB.  gfx950
correction (same guide, 'HBM'): FETCH_SIZE reports exactly half of a coalesced streaming read -> doubled; WRITE_SIZE
is exact.  Values are per launch of the named kernel (mean over the profiled launches)."""
import csv, glob, json, sys
fetch_dir, write_dir, kernel, steps_per_launch, envs, out = sys.argv[1:7]

def mean_counter(d, name):
    vals = []
    for f in (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)

fk, nf = mean_counter(fetch_dir, "FETCH_SIZE")
wk, nw = mean_counter(write_dir, "WRITE_SIZE")
res = dict(kernel=kernel, steps_per_launch=int(steps_per_launch), envs=int(envs), launches_profiled=[nf, nw],
           FETCH_SIZE_KiB=fk, WRITE_SIZE_KiB=wk, fetch_correction=2.0,
           hbm_bytes_per_launch=(2.0 * fk + wk) * 1024.0,
           hbm_bytes_per_step=(2.0 * fk + wk) * 1024.0 / int(steps_per_launch))
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
