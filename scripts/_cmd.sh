cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu -k "render or frame_stack or birdview or obs or config1 or loader" 2>&1 | tail -3
python scripts/ab_render.py ab/libpaint1.so torchdriveenv_amd/libtde_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_render_carry.txt
