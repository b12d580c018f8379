#!/bin/bash
# Round-4 measurement pass on the GPU box: gpurun -- bash scripts/measure_r04.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r04m}
exec < /dev/null
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
python bench.py --world town --no-secondary --no-cpu-baseline > $O/bench_town.json 2>> $O/bench.err
python bench.py --world town --config 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_town_config5.json 2>> $O/bench.err
python bench.py --world town --config 5 --streams 1 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_town_config5_1stream.json 2>> $O/bench.err
python bench.py --world town --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_town_stepmode.json 2>> $O/bench.err
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
python bench.py --config 5 --streams 1 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5_1stream.json 2>> $O/bench.err
python bench.py --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode.json 2>> $O/bench.err
L=torchdriveenv_amd/libtde_hip.so
python scripts/ab_render.py $L > $O/render.txt 2>/dev/null; python scripts/ab_render.py --town $L >> $O/render.txt 2>/dev/null
grep -v amdgpu $O/render.txt
# kernel stats: the headline command, and the town
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_town -o st -- python3 bench.py --world town --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats_town.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_town_config5 -o st -- python3 bench.py --world town --config 5 --streams 1 --steps 300 --warmup 30 --no-cpu-baseline > $O/stats_town5.log 2>&1
sleep 2
# HBM traffic (separate passes: FETCH_SIZE and WRITE_SIZE cannot share one), junction maps and town, rollout and rasteriser
for W in junctions town; do
  for c in FETCH_SIZE WRITE_SIZE; do
    TDE_WORLD=$W timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${W}_$c -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 3 > $O/pmc_${W}_$c.log 2>&1; sleep 1
    TDE_WORLD=$W timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/render_${W}_$c -o pmc --output-format csv -- python3 scripts/run_render.py 10 > $O/render_${W}_$c.log 2>&1; sleep 1
    TDE_WORLD=$W timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/step_${W}_$c -o pmc --output-format csv -- python3 scripts/run_step.py 300 > $O/step_${W}_$c.log 2>&1; sleep 1
  done
  python scripts/traffic_from_pmc.py $O/pmc_${W}_FETCH_SIZE $O/pmc_${W}_WRITE_SIZE env_rollout_trio_kernel 250 8192 $O/traffic_rollout_$W.json
  python scripts/traffic_from_pmc.py $O/render_${W}_FETCH_SIZE $O/render_${W}_WRITE_SIZE render_views_kernel 1 8192 $O/traffic_render_$W.json
  python scripts/traffic_from_pmc.py $O/step_${W}_FETCH_SIZE $O/step_${W}_WRITE_SIZE env_step_trio_kernel 1 8192 $O/traffic_step_$W.json
done
ls $O
# closed loop through a HIP graph (no host in the loop) with / without re-spawns, stamps of the three-role step kernel
python scripts/ab_step.py $L > $O/ab_step.txt 2>&1; python scripts/ab_step.py --endless $L >> $O/ab_step.txt 2>&1
python scripts/ab_step.py --town $L >> $O/ab_step.txt 2>&1; python scripts/ab_step.py --outputs $L >> $O/ab_step.txt 2>&1
python scripts/ab_step.py --envs 1024 $L >> $O/ab_step.txt 2>&1; python scripts/ab_step.py --envs 4096 $L >> $O/ab_step.txt 2>&1
grep -v amdgpu $O/ab_step.txt
if [ -f ab/libS.so ]; then TDE_HIP_LIB=$PWD/ab/libS.so python scripts/step_stamps.py > $O/step_stamps.txt 2>&1; grep -v amdgpu $O/step_stamps.txt; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -o st -- python3 bench.py --mode step --steps 2000 --warmup 200 --no-cpu-baseline > $O/stats_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5 -o st -- python3 bench.py --config 5 --streams 1 --steps 300 --warmup 30 --no-cpu-baseline > $O/stats5.log 2>&1
# SQ counters of the headline rollout per 64-slot group and step (/ 512 000 = 2048 groups x 250 steps)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 2 > $O/pmc_sq.log 2>&1
python scripts/pmc_summary.py --div=512000 $O/pmc_sq | grep -A9 env_rollout_trio > $O/pmc_sq_per_group_step.txt 2>&1; cat $O/pmc_sq_per_group_step.txt
