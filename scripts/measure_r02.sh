#!/bin/bash
# Round-2 measurement pass on the GPU box: gpurun -- bash scripts/measure_r02.sh <tag>
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02b}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_shape.json 2>> $O/bench.err
python bench.py --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode.json 2>> $O/bench.err
python bench.py --mode step --binding ctypes --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode_ctypes.json 2>> $O/bench.err
python bench.py --mode step --envs 1024 --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode_1024envs.json 2>> $O/bench.err
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
python bench.py --config 2 --steps 20000 --warmup 500 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
python bench.py --gpus 2 --backend gloo --envs 4096 --steps 2000 --warmup 250 > $O/bench_2ranks_1gpu_gloo.json 2>> $O/bench.err
python scripts/bench_config2.py > $O/config2_per_call.json 2>/dev/null
python scripts/bench_vecenv.py > $O/python_boundary.json 2>/dev/null
python scripts/launch_cost.py > $O/launch_cost.txt 2>/dev/null; TDE_STEP=solo python scripts/launch_cost.py >> $O/launch_cost.txt 2>/dev/null; TDE_STEP=trio python scripts/launch_cost.py >> $O/launch_cost.txt 2>/dev/null
# interleaved same-process A/B: round-1 library vs this build; duo vs trio per group shape (two copies of the library: the
# TDE_ROLLOUT choice is latched per loaded library)
cp torchdriveenv_amd/libtde_hip.so ab/libcur_duo.so; cp torchdriveenv_amd/libtde_hip.so ab/libcur_trio.so
# (the round-1 library cannot be loaded next to this build any more: tde_state / tde_map changed with ABI 5 and 6; the
#  recorded A/B is profiles/r02_a_ab_r01_vs_r02.txt, later steps are A/B-ed build against build in profiles/r02_d_*.txt)
: > $O/rollout_matrix.txt
for shape in "8 16384" "16 8192" "32 4096" "64 2048"; do set -- $shape
  python scripts/ab_rollout.py --agents $1 --envs $2 duo:ab/libcur_duo.so trio:ab/libcur_trio.so >> $O/rollout_matrix.txt 2>/dev/null
  python scripts/ab_rollout.py --agents $1 --envs $2 --lights duo:ab/libcur_duo.so trio:ab/libcur_trio.so >> $O/rollout_matrix.txt 2>/dev/null
done
cat $O/rollout_matrix.txt
python scripts/ab_render.py torchdriveenv_amd/libtde_hip.so > $O/render.txt 2>/dev/null; python scripts/ab_render.py --stack 3 torchdriveenv_amd/libtde_hip.so >> $O/render.txt 2>/dev/null; python scripts/ab_render.py --lights torchdriveenv_amd/libtde_hip.so >> $O/render.txt 2>/dev/null
python scripts/ablate.py > $O/ablation.txt 2>/dev/null
python scripts/scale_envs.py > $O/scale_envs.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config5 -o st -- python3 bench.py --config 5 --steps 300 --warmup 30 --no-cpu-baseline > $O/stats5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_step -o st -- python3 bench.py --mode step --steps 2000 --warmup 200 --no-cpu-baseline > $O/stats_step.log 2>&1
sleep 3
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 3 > $O/pmc_fetch.log 2>&1
sleep 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 3 > $O/pmc_write.log 2>&1
sleep 3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 2 > $O/pmc_sq.log 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write env_rollout_trio_kernel 250 8192 $O/traffic.json
python scripts/pmc_summary.py $O/pmc_sq env_rollout_trio_kernel > $O/pmc_sq_per_group_step.txt 2>&1; cat $O/pmc_sq_per_group_step.txt
ls $O
