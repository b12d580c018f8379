set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
python bench.py --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode.json 2>> $O/bench.err
python bench.py --rollout-kernel solo --no-cpu-baseline > $O/bench_solo.json 2>> $O/bench.err
python bench.py --rollout-kernel duo --no-cpu-baseline > $O/bench_duo.json 2>> $O/bench.err
python scripts/rollout_matrix.py > $O/rollout_matrix_default.txt 2>/dev/null
TDE_ROLLOUT=duo python scripts/rollout_matrix.py > $O/rollout_matrix_duo.txt 2>/dev/null
python scripts/ablate.py > $O/ablation.txt 2>/dev/null
python scripts/scale_envs.py > $O/scale_envs.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --steps 10000 --warmup 1000 --no-cpu-baseline > $O/stats.log 2>&1
sleep 5
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 3 > $O/pmc_fetch.log 2>&1
sleep 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 3 > $O/pmc_write.log 2>&1
sleep 3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU -d $O/pmc_sq -o pmc --output-format csv -- python3 scripts/run_rollout.py 31 2 > $O/pmc_sq.log 2>&1
ls $O $O/stats | head -40
