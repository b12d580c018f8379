cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "step or env or cache or fuzz or offroad" 2>&1 | tail -3
python bench.py --config 5 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5', d['ms_per_step']*1e3, d['roofline']['frac'])"
python bench.py --mode step --step-kernel solo --steps 4000 --warmup 500 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step solo', d['ms_per_step']*1e3)"
