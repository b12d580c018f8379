cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python scripts/ab_rollout.py ab/libpre_bm.so torchdriveenv_amd/libtde_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_ab_boxmuller.txt
