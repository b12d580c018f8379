cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for B in 8192 12288 16384 20480 32768 65536; do python bench.py --envs $B --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($B, round(d['ms_per_step']*1e3,3), 'us/step', '%.3e' % d['agent_steps_per_sec'], 'kernel_avg_us', round(d['roofline']['kernel_avg_us'],1))"; done | tee gpurun_out/r03_scale_envs_chunked.txt
