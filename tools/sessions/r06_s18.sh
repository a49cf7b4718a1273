#!/bin/bash
# round 6, GPU session 18: composite + mean image loss + their adjoints in one launch: its parity test, the whole GPU suite, one-view and 8-view lines against the separate kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s18; O=$R/gpurun_out/r6s18
echo "== fused loss test"; timeout 600 python -m pytest tests/test_gpu_renderutils.py -q -m gpu -x -k fused_composite 2>&1 | grep -v Warning | tail -15 | tee $O/test_fused.txt
echo "== gpu tests"; timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -4 | tee $O/tests.txt
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}; r=d['roofline']
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), {k: e.get(k) for k in ('mode','exposed_ms','geometry_stage_ms')} if e else '')" || tail -3 $O/$name.err | cut -c1-200; }
for rep in 1 2; do
one bob_1view_$rep --batch 1 --graph on
NVDR_TUNING=1 NVDR_FUSED_LOSS=0 one bob_1view_separate_$rep --batch 1 --graph on
one bob_1view_schedule_$rep --batch 1 --graph on --exchange-world1
NVDR_TUNING=1 NVDR_FUSED_LOSS=0 one bob_1view_schedule_separate_$rep --batch 1 --graph on --exchange-world1
done
one dmtet800_1view_trained --config dmtet800 --batch 1 --graph on
NVDR_TUNING=1 NVDR_FUSED_LOSS=0 one dmtet800_1view_trained_separate --config dmtet800 --batch 1 --graph on
one dmtet800_1view_locked_schedule --config dmtet800 --lock-pos on --batch 1 --graph on --exchange-world1
NVDR_TUNING=1 NVDR_FUSED_LOSS=0 one dmtet800_1view_locked_schedule_separate --config dmtet800 --lock-pos on --batch 1 --graph on --exchange-world1
one bob_8views --steps 50 --warmup 10
NVDR_TUNING=1 NVDR_FUSED_LOSS=0 one bob_8views_separate --steps 50 --warmup 10
