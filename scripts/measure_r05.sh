#!/bin/bash
# Round-5 measurement pass on the GPU box: gpurun -- bash scripts/measure_r05.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r05m}
exec < /dev/null
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
python bench.py --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode.json 2>> $O/bench.err
python bench.py --world town --no-secondary --no-cpu-baseline > $O/bench_town.json 2>> $O/bench.err
# the one-launch step with / without the magnitudes, same process, interleaved
python scripts/step_magnitudes_cost.py junctions town wide a32 > $O/step_magnitudes_cost.txt 2>&1; grep -v amdgpu $O/step_magnitudes_cost.txt
# kernel stats: the headline command; the closed loop with full outputs + magnitudes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
TDE_STEP_OUTPUTS=mag rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step_mag -o st -- python3 scripts/run_step.py 2000 > $O/stats_step_mag.log 2>&1
TDE_STEP_OUTPUTS=full rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step_full -o st -- python3 scripts/run_step.py 2000 > $O/stats_step_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step_bare -o st -- python3 scripts/run_step.py 2000 > $O/stats_step_bare.log 2>&1
sleep 2
# HBM traffic of the closed-loop step with magnitudes (separate passes: FETCH_SIZE and WRITE_SIZE cannot share one), and of the rollout
for c in FETCH_SIZE WRITE_SIZE; do
  TDE_STEP_OUTPUTS=mag timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/stepmag_$c -o pmc --output-format csv -- python3 scripts/run_step.py 300 > $O/stepmag_$c.log 2>&1; sleep 1
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 3 > $O/pmc_$c.log 2>&1; sleep 1
done
python scripts/traffic_from_pmc.py $O/stepmag_FETCH_SIZE $O/stepmag_WRITE_SIZE env_step_trio_kernel 1 8192 $O/traffic_step_mag.json
python scripts/traffic_from_pmc.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE env_rollout_trio_kernel 250 8192 $O/traffic_rollout.json
# python boundary / VecEnv
python scripts/bench_vecenv.py > $O/vecenv.txt 2>&1; grep -v amdgpu $O/vecenv.txt | tail -12
find $O -name "*.csv" -size +3M -delete
ls $O
