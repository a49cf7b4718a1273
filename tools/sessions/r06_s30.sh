#!/bin/bash
# round 6, GPU session 30: twenty more repeats of the trained one-view schedule line (stderr kept), looking for the run that ended without output in the measurement set
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s30; O=$R/gpurun_out/r6s30
bad=0
for rep in $(seq 1 20); do
  SECONDS=0
  timeout 120 python bench.py --config dmtet800 --batch 1 --graph on --exchange-world1 --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 60 --warmup 20 2>$O/err.txt | tail -1 > $O/out.json
  rc=$?
  if python -c "
import json; d=json.load(open('$O/out.json')); print('rep $rep', round(d['ms_per_step'],3), d['steps_over_twice_the_median'], 'wall ${SECONDS}s')" 2>/dev/null; then :; else
    bad=$((bad+1)); cp $O/err.txt $O/err_fail_$rep.txt; echo "rep $rep FAILED rc=$rc wall ${SECONDS}s"; grep -v "Warning\|warn" $O/err.txt | tail -25 | cut -c1-300
  fi
done
echo "failures: $bad of 20"
