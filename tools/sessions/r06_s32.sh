#!/bin/bash
# round 6, GPU session 32: the sporadic end of schedule runs explained (the process group's watchdog thread queries events while the harness captures: an error in 'global' capture mode) and fixed (thread-local capture mode): the probe, the tests, 24 repeats of the two lines that had failed, then the measurement set
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s32; O=$R/gpurun_out/r6s32
timeout 120 python tools/capture_mode_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/capture_mode_probe.txt
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_geometry.py -q -m gpu -x 2>&1 | grep -v Warning | tail -3 | tee $O/tests.txt
bad=0
for rep in $(seq 1 12); do
  for kind in trained sparse; do
    if [ $kind = trained ]; then A="--config dmtet800 --batch 1 --graph on --exchange-world1"; else A="--batch 1 --graph on --exchange-world1 --exchange sparse"; fi
    timeout 120 python bench.py $A --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 40 --warmup 20 2>$O/err.txt | tail -1 > $O/out.json
    if python -c "
import json; d=json.load(open('$O/out.json')); print('rep $rep $kind', round(d['ms_per_step'],3), d['steps_over_twice_the_median'])" 2>/dev/null; then :; else
      bad=$((bad+1)); cp $O/err.txt $O/err_fail_${rep}_$kind.txt; echo "rep $rep $kind FAILED"; grep -v "Warning\|warn\|frame #" $O/err.txt | tail -8 | cut -c1-300
    fi
  done
done 2>&1 | tee $O/repeats.txt
echo "failures: $bad of 24" | tee -a $O/repeats.txt
bash tools/r06_final.sh
