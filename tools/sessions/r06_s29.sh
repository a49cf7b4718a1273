#!/bin/bash
# round 6, GPU session 29: the trained one-view schedule line came back empty once in the measurement set: repeat it with its stderr kept
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s29; O=$R/gpurun_out/r6s29
for rep in 1 2 3 4 5 6; do
  SECONDS=0
  timeout 200 python bench.py --config dmtet800 --batch 1 --graph on --exchange-world1 --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/err_$rep.txt | tail -1 > $O/out_$rep.json
  echo "rep $rep rc=$? wall ${SECONDS}s bytes $(stat -c %s $O/out_$rep.json)"; python -c "
import json; d=json.load(open('$O/out_$rep.json')); e=d['config'].get('exchange') or {}
print(round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['steps_over_twice_the_median'], e.get('geometry_stage_ms'))" 2>/dev/null || { echo "--- stderr tail"; grep -v "Warning\|warn" $O/err_$rep.txt | tail -15 | cut -c1-300; }
done
