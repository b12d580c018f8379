#!/bin/bash
# Round-3 baseline pass on the GPU box: gpurun -- bash scripts/r03_baseline.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03_base}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 900 $O/bench.json
python bench.py --mode step --steps 4000 --warmup 500 --no-cpu-baseline > $O/bench_stepmode.json 2>> $O/bench.err; tail -c 600 $O/bench_stepmode.json
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/bench_config5.json 2>> $O/bench.err; tail -c 600 $O/bench_config5.json
python scripts/ab_render.py torchdriveenv_amd/libtde_hip.so > $O/render.txt 2>/dev/null; cat $O/render.txt
python scripts/scale_envs.py > $O/scale_envs.txt 2>/dev/null; cat $O/scale_envs.txt
