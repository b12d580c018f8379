cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_step_flags3; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/solo32 -o tr -- python3 scripts/step_flags_rocprof.py 32 8192 solo > $O/solo32.log 2>&1
python - <<PY
import csv, glob
rows = [r for f in glob.glob("$O/solo32/*kernel_trace.csv") for r in csv.DictReader(open(f)) if "env_step" in r["Kernel_Name"]]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
for s in range(5):
    seg = sorted(d[s*600+100:(s+1)*600])
    print("solo A=32 subset", s, "kernel avg %.2f us  median %.2f  p10 %.2f p90 %.2f" % (sum(seg)/len(seg)/1e3, seg[len(seg)//2]/1e3, seg[len(seg)//10]/1e3, seg[9*len(seg)//10]/1e3))
PY
