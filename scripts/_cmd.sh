cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 1024 4096 8192; do TDE_HIP_LIB=$PWD/ab/libS.so python scripts/phase_stamps.py trio $B 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r03_phase_stamps_trio.txt
