#!/bin/bash
# Round-6 GPU pass: gpurun -- bash scripts/gpu_r06.sh <tag> [tests] [ab:<libA>:<libB>...] [ubench]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
exec < /dev/null
TAG=${1:-r06}; shift
O=gpurun_out/$TAG; mkdir -p $O
for what in "$@"; do
  case "$what" in
    ubench) ./scripts/ubench/shift64_last_vgpr > $O/shift64_last_vgpr.txt 2>&1; tail -12 $O/shift64_last_vgpr.txt;;
    tests) ( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.txt 2>&1; tail -4 $O/pytest_gpu.txt;;
    ab:*) LIBS=$(echo "${what#ab:}" | tr ':' ' ')
      python scripts/ab_rollout.py $LIBS > $O/ab_rollout.txt 2>&1; grep -v amdgpu.ids $O/ab_rollout.txt | tail -4
      python scripts/ab_rollout.py --lights $LIBS > $O/ab_rollout_lights.txt 2>&1; grep -v amdgpu.ids $O/ab_rollout_lights.txt | tail -4
      python scripts/ab_step.py $LIBS > $O/ab_step.txt 2>&1; grep -v amdgpu.ids $O/ab_step.txt | tail -4
      python scripts/ab_step.py --lights --outputs $LIBS > $O/ab_step_lights.txt 2>&1; grep -v amdgpu.ids $O/ab_step_lights.txt | tail -4;;
    abcoast:*) LIBS=$(echo "${what#abcoast:}" | tr ':' ' ')
      for opt in "" "--coast" "--lights" "--lights --coast"; do
        echo "== rollout $opt" >> $O/ab_first_step.txt; python scripts/ab_rollout.py $opt $LIBS 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/ab_first_step.txt
        echo "== step $opt" >> $O/ab_first_step.txt; python scripts/ab_step.py $opt $LIBS 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/ab_first_step.txt
      done; cat $O/ab_first_step.txt;;
    t:*) ( time timeout 1200 python -m pytest tests -m gpu -q -x -k "${what#t:}" ) > $O/pytest_k.txt 2>&1; tail -15 $O/pytest_k.txt;;
    abstep:*) LIBS=$(echo "${what#abstep:}" | tr ':' ' ')
      for opt in "--coast" ""; do
        echo "== step $opt" >> $O/ab_step_bisect.txt; python scripts/ab_step.py $opt $LIBS 2>&1 | grep -v amdgpu.ids | tail -5 >> $O/ab_step_bisect.txt
      done; cat $O/ab_step_bisect.txt;;
    prof:*) LIBS=$(echo "${what#prof:}" | tr ':' ' ')
      for L in $LIBS; do for C in 1 0; do
        T=$(basename $L .so)_coast$C
        TDE_HIP_LIB=$PWD/$L TDE_COAST=$C rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -o st -- python3 scripts/run_step.py 2000 > $O/prof_$T.log 2>&1
        echo "== $T"; grep -h "env_step_trio\|first_gap" $O/prof_$T/*/st_kernel_stats.csv 2>/dev/null | cut -c1-200 || find $O/prof_$T -name "*stats*" | head
      done; done; find $O -name "*.csv" -size +2M -delete;;
    ablights:*) LIBS=$(echo "${what#ablights:}" | tr ':' ' ')
      python scripts/ab_rollout.py --lights $LIBS 2>&1 | grep -v amdgpu.ids | tail -5 > $O/ab_lights.txt
      python scripts/ab_step.py --lights $LIBS 2>&1 | grep -v amdgpu.ids | tail -5 >> $O/ab_lights.txt
      python scripts/ab_rollout.py --lights --town --signals 64 $LIBS 2>&1 | grep -v amdgpu.ids | tail -5 >> $O/ab_lights.txt
      cat $O/ab_lights.txt;;
    wide) python scripts/wide_times.py > $O/wide_times.txt 2>&1; grep -v amdgpu $O/wide_times.txt;;
    wideforms) python scripts/wide_step_forms.py > $O/wide_step_forms.txt 2>&1; grep -v amdgpu $O/wide_step_forms.txt;;
    stamps) for L in ab/libS_nofg.so ab/libS_fg.so; do for opt in "--coast" ""; do echo "== $L $opt" >> $O/step_stamps.txt; TDE_HIP_LIB=$PWD/$L python scripts/step_stamps.py $opt 2>&1 | grep -v amdgpu >> $O/step_stamps.txt; done; done; cat $O/step_stamps.txt;;
    abprio:*) LIBS=$(echo "${what#abprio:}" | tr ':' ' ')
      for opt in "" "--outputs" "--lights"; do echo "== step $opt" >> $O/ab_step_prio.txt; python scripts/ab_step.py $opt $LIBS 2>&1 | grep -v amdgpu.ids | tail -7 >> $O/ab_step_prio.txt; done; cat $O/ab_step_prio.txt;;
    robust) ( time TDE_FUZZ_CASES=120 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -q -n 8 ) > $O/fuzz360.txt 2>&1; tail -3 $O/fuzz360.txt
      timeout 900 python scripts/soak.py > $O/soak.txt 2>&1; grep -v amdgpu $O/soak.txt | tail -8
      timeout 900 python scripts/stress_step_forms.py 60 > $O/stress_step_forms.txt 2>&1; grep -v amdgpu $O/stress_step_forms.txt | tail -6;;
    abdark:*) LIBS=$(echo "${what#abdark:}" | tr ':' ' ')
      for opt in "" "--lights --dark-world" "--lights"; do echo "== rollout $opt" >> $O/ab_dark.txt; python scripts/ab_rollout.py $opt $LIBS 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/ab_dark.txt; done; cat $O/ab_dark.txt;;
    smoke) ( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt;;
    durations) ( time timeout 1800 python -m pytest tests -m gpu -q --durations=25 ) > $O/pytest_durations.txt 2>&1; tail -40 $O/pytest_durations.txt;;
    bench) python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json;;
    *) echo "unknown phase $what";;
  esac
done
