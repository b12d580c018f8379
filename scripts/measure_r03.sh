#!/bin/bash
# Round-3 measurement pass on the GPU box: gpurun -- bash scripts/measure_r03.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03m}
exec < /dev/null
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_driver_shape.json 2>> $O/bench.err
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
python bench.py --config 5 --streams 1 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5_1stream.json 2>> $O/bench.err
python bench.py --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode.json 2>> $O/bench.err
python bench.py --mode step --step-kernel trio --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode_trio.json 2>> $O/bench.err
python bench.py --config 2 --steps 20000 --warmup 500 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
python scripts/bench_vecenv.py > $O/python_boundary.json 2>/dev/null
L=torchdriveenv_amd/libtde_hip.so
python scripts/ab_render.py $L > $O/render.txt 2>/dev/null; python scripts/ab_render.py --stack 3 $L >> $O/render.txt 2>/dev/null; python scripts/ab_render.py --lights $L >> $O/render.txt 2>/dev/null; python scripts/ab_render.py --agents 16 $L >> $O/render.txt 2>/dev/null
grep -v amdgpu $O/render.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5 -o st -- python3 bench.py --config 5 --streams 1 --steps 300 --warmup 30 --no-cpu-baseline > $O/stats5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5_streams -o st -- python3 bench.py --config 5 --steps 300 --warmup 30 --no-cpu-baseline > $O/stats5s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -o st -- python3 bench.py --mode step --steps 2000 --warmup 200 --no-cpu-baseline > $O/stats_step.log 2>&1
sleep 2
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 3 > $O/pmc_fetch.log 2>&1
sleep 2
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 3 > $O/pmc_write.log 2>&1
sleep 2
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 2 > $O/pmc_sq.log 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write env_rollout_trio_kernel 250 8192 $O/traffic.json
python scripts/pmc_summary.py --div=512000 $O/pmc_sq | grep -A9 env_rollout_trio > $O/pmc_sq_per_group_step.txt 2>&1; cat $O/pmc_sq_per_group_step.txt
# rasteriser: traffic and SQ / TA counters per view
for c in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/render_$c -o pmc --output-format csv -- python3 scripts/run_render.py 10 > $O/render_$c.log 2>&1; sleep 1; done
python scripts/traffic_from_pmc.py $O/render_FETCH_SIZE $O/render_WRITE_SIZE render_views_kernel 1 8192 $O/render_traffic.json
TAG=$TAG/render_pmc bash scripts/r03_render_pmc.sh > /dev/null 2>&1; cat gpurun_out/$TAG/render_pmc/per_view.txt
# calibration of the FETCH_SIZE correction on a plain 256 MiB device-to-device copy
cat > /tmp/copy256.py <<'PY'
import torch
x = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device="cuda:0"); y = torch.empty_like(x)
for _ in range(5): y.copy_(x)
torch.cuda.synchronize()
PY
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/copy_fetch -o pmc --output-format csv -- python3 /tmp/copy256.py > /dev/null 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/copy_write -o pmc --output-format csv -- python3 /tmp/copy256.py > /dev/null 2>&1
python - <<PY > $O/copy_calibration.txt
import csv, glob
for d, c in (("$O/copy_fetch", "FETCH_SIZE"), ("$O/copy_write", "WRITE_SIZE")):
    v = [float(r["Counter_Value"]) for f in glob.glob(d + "/*counter_collection.csv") for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "copy" in r["Kernel_Name"].lower()]
    print(c, "KiB per 256 MiB copy kernel:", sorted(v)[len(v) // 2] if v else None, "(262144 KiB moved each way)")
PY
cat $O/copy_calibration.txt
ls $O
