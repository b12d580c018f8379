cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python scripts/ab_render.py ab/libvpw8.so torchdriveenv_amd/libtde_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_render_qorder.txt
