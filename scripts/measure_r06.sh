#!/bin/bash
# Round-6 measurement pass on the GPU box: gpurun -- bash scripts/measure_r06.sh <tag> [notests]
# Writes gpurun_out/<tag>/: the GPU suite's log, bench lines, rocprofv3 kernel stats, the PMC traffic passes and - from THIS pass -
# traffic.json / traffic_config5.json in the form bench.py reads from profiles/ (copy them there: the note names the pass).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r06m}
exec < /dev/null
O=gpurun_out/$TAG; mkdir -p $O
FL=159        # TDE_F_ALL: NPC | REPLAY | OFFROAD | REWARD | AUTORESET | NPC_FIRST_STEP
if [ "$2" != "notests" ]; then ( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt; fi
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
python bench.py --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode.json 2>> $O/bench.err
python bench.py --world town --no-secondary --no-cpu-baseline > $O/bench_town.json 2>> $O/bench.err
# kernel stats: the headline command; the closed loop bare / with every output + magnitudes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
TDE_STEP_OUTPUTS=mag rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step_mag -o st -- python3 scripts/run_step.py 2000 > $O/stats_step_mag.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step_bare -o st -- python3 scripts/run_step.py 2000 > $O/stats_step_bare.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5 -o st -- python3 bench.py --config 5 --streams 1 --steps 300 --warmup 30 --no-cpu-baseline > $O/stats5.log 2>&1
sleep 2
# HBM traffic (separate passes: FETCH_SIZE and WRITE_SIZE cannot share one): the rollout, the closed-loop step with magnitudes, and the
# two kernels of configs[4] (rasteriser + the 32-slot one-role step)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python3 scripts/run_rollout.py $FL 3 > $O/pmc_$c.log 2>&1; sleep 1
  TDE_STEP_OUTPUTS=mag timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/stepmag_$c -o pmc --output-format csv -- python3 scripts/run_step.py 300 > $O/stepmag_$c.log 2>&1; sleep 1
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/render_$c -o pmc --output-format csv -- python3 scripts/run_render.py 10 > $O/render_$c.log 2>&1; sleep 1
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/step32_$c -o pmc --output-format csv -- python3 scripts/run_step.py 200 solo 32 8192 > $O/step32_$c.log 2>&1; sleep 1
done
python scripts/traffic_from_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE env_rollout_trio_kernel 250 8192 $O/traffic_rollout.json
python scripts/traffic_from_pmc.py $O/stepmag_FETCH_SIZE $O/stepmag_WRITE_SIZE env_step_trio_kernel 1 8192 $O/traffic_step_mag.json
python scripts/traffic_from_pmc.py $O/render_FETCH_SIZE $O/render_WRITE_SIZE render_views_kernel 1 8192 $O/traffic_render.json
python scripts/traffic_from_pmc.py $O/step32_FETCH_SIZE $O/step32_WRITE_SIZE env_step_kernel 1 8192 $O/traffic_step32.json
# SQ counters of the headline rollout per 64-slot group and step (/ 512 000 = 2048 groups x 250 steps)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc --output-format csv -- python3 scripts/run_rollout.py $FL 2 > $O/pmc_sq.log 2>&1
python scripts/pmc_summary.py --div=512000 $O/pmc_sq | grep -A9 env_rollout_trio > $O/pmc_sq_per_group_step.txt 2>&1; cat $O/pmc_sq_per_group_step.txt
# ... and what bench.py reads from profiles/: traffic.json / traffic_config5.json of THIS pass
python scripts/finish_traffic.py $O $TAG
find $O -name "*.csv" -size +3M -delete
ls $O
