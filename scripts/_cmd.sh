cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5 default', d['ms_per_step']*1e3, d['roofline']['frac'])"
TDE_STEP=trio python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5 trio', d['ms_per_step']*1e3, d['roofline']['frac'])"
