cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu -k "render or frame_stack or birdview or obs or config1" 2>&1 | tail -3
python scripts/ab_render.py ab/libpre_mlp.so torchdriveenv_amd/libtde_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_render_mlp2.txt
python scripts/ab_render.py --lights torchdriveenv_amd/libtde_hip.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_render_mlp2.txt
