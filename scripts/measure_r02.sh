#!/bin/bash
# Round-2 measurement pass on the GPU box: gpurun -- bash scripts/measure_r02.sh <tag> [quick]
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r02a}
O=gpurun_out/$TAG; mkdir -p $O
if [ "$2" != "quick" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
fi
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 900 $O/bench.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_shape.json 2>> $O/bench.err; tail -c 700 $O/bench_driver_shape.json
python bench.py --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode.json 2>> $O/bench.err
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err
python bench.py --config 2 --steps 20000 --warmup 500 --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
TDE_HIP_LIB=$PWD/ab/libS.so python scripts/phase_stamps.py trio 8192 > $O/phase_stamps_trio.txt 2>&1
TDE_HIP_LIB=$PWD/ab/libS.so python scripts/phase_stamps.py trio 1024 >> $O/phase_stamps_trio.txt 2>&1
cat $O/phase_stamps_trio.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
ls $O $O/stats | head -30
