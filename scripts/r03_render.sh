#!/bin/bash
# rasteriser pass on the GPU box: pixel parity of the in-tree build, then interleaved A/B against ab/*.so given as args
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r03_render}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "render or frame_stack or birdview or obs or config1" > $O/pytest_render.txt 2>&1; tail -5 $O/pytest_render.txt
python scripts/ab_render.py "$@" > $O/ab_render.txt 2>&1
python scripts/ab_render.py --stack 3 "$@" >> $O/ab_render.txt 2>&1; python scripts/ab_render.py --lights "$@" >> $O/ab_render.txt 2>&1
python scripts/ab_render.py --agents 16 "$@" >> $O/ab_render.txt 2>&1
grep -v amdgpu.ids $O/ab_render.txt | tail -16
L=torchdriveenv_amd/libtde_hip.so
: > $O/scale_views.txt
for B in 256 1024 4096 8192 16384; do python scripts/ab_render.py --envs $B --launches 20 $L 2>/dev/null | tail -1 | sed "s/^/B=$B /" >> $O/scale_views.txt; done
cat $O/scale_views.txt
