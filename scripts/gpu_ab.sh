#!/bin/bash
# quick GPU pass while tuning kernels: parity of the in-tree build, then interleaved A/B of ab/*.so given as args
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-ab}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -x -q -k "not bench" > $O/pytest_parity.txt 2>&1; tail -3 $O/pytest_parity.txt
python scripts/ab_rollout.py "$@" > $O/ab.txt 2>&1; cat $O/ab.txt
if [ -f ab/libS.so ]; then
  TDE_HIP_LIB=$PWD/ab/libS.so python scripts/phase_stamps.py trio 8192 > $O/phase_stamps_trio.txt 2>&1
  TDE_HIP_LIB=$PWD/ab/libS.so python scripts/phase_stamps.py trio 1024 >> $O/phase_stamps_trio.txt 2>&1
  cat $O/phase_stamps_trio.txt
fi
