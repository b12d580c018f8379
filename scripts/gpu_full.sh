#!/bin/bash
# full GPU pass: the whole -m gpu suite on the in-tree build, then interleaved A/B of the libraries given as arguments
# on the headline rollout and on the rasteriser (configs[4])
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-full}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python scripts/ab_rollout.py "$@" > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
python scripts/ab_render.py "$@" > $O/ab_render.txt 2>&1; grep -v amdgpu.ids $O/ab_render.txt
