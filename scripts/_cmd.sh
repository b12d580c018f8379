cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python scripts/ab_render.py torchdriveenv_amd/libtde_hip.so ab/libskip1.so ab/libskip2.so ab/libskip4.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_render_ablate4.txt
python scripts/ab_render.py --agents 16 torchdriveenv_amd/libtde_hip.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_render_ablate4.txt
python scripts/ab_render.py --lights torchdriveenv_amd/libtde_hip.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_render_ablate4.txt
python scripts/ab_render.py --stack 3 torchdriveenv_amd/libtde_hip.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03_render_ablate4.txt
