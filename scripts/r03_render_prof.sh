#!/bin/bash
# profile of the rasteriser: scaling with the number of views, SQ counters, HBM traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${TAG:-r03_render_prof}
O=gpurun_out/$TAG; mkdir -p $O
L=torchdriveenv_amd/libtde_hip.so
: > $O/scale_views.txt
for B in 256 1024 2048 4096 8192 16384; do python scripts/ab_render.py --envs $B --launches 20 $L 2>/dev/null | tail -1 | sed "s/^/B=$B /" >> $O/scale_views.txt; done
cat $O/scale_views.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 scripts/run_render.py 50 > $O/stats.log 2>&1
grep render $O/stats/*kernel_stats.csv | head -3
sleep 2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc --output-format csv -- python3 scripts/run_render.py 10 > $O/pmc_sq.log 2>&1
python scripts/pmc_summary.py --div=8192 $O/pmc_sq | grep -A9 render_views > $O/pmc_sq_per_view.txt; cat $O/pmc_sq_per_view.txt
sleep 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o pmc --output-format csv -- python3 scripts/run_render.py 10 > $O/pmc_fetch.log 2>&1
sleep 2
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o pmc --output-format csv -- python3 scripts/run_render.py 10 > $O/pmc_write.log 2>&1
python scripts/traffic_from_pmc.py $O/pmc_fetch $O/pmc_write render_views_kernel 1 8192 $O/render_traffic.json
