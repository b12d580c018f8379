#!/bin/bash
# SQ / TA / TCP counters of the rasteriser per view (separate passes; the TA_FLAT_* / TA_*_STALLED_* sets hang on this pool)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r03_render_pmc}
O=gpurun_out/$TAG; mkdir -p $O
: > $O/per_view.txt
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | cut -c1-14 | tr ' ' '_')
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $O/pmc_$tag -o pmc --output-format csv -- python3 scripts/run_render.py 6 > $O/pmc_$tag.log 2>&1
  python scripts/pmc_summary.py --div=8192 $O/pmc_$tag | grep -A9 render_views >> $O/per_view.txt
  sleep 1
done
cat $O/per_view.txt
