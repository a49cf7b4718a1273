#!/bin/bash
# round 6, GPU session 28: a refit that keeps the eight-wide collapse of the last full build (boxes, slot order and nodes redone; no dynamic programme, budgets or prefix sums): parity, refit times, one-view lines of the trained 684 k mesh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s28; O=$R/gpurun_out/r6s28
echo "== tests"; timeout 1800 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_geometry.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | grep -v Warning | tail -3 | tee $O/tests.txt
for sd in 3 0; do timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep triangles; NVDR_TUNING=1 NVDR_REFIT_KEEP_COLLAPSE=0 timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep "refit" | sed 's/$/  [NVDR_REFIT_KEEP_COLLAPSE=0]/'; done | tee $O/bvh_build_alone.txt
timeout 200 python tools/bvh_probe.py dmtet64_mid 0 40 2>&1 | grep triangles | tee -a $O/bvh_build_alone.txt
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), {k: e.get(k) for k in ('mode','exposed_ms','geometry_stage_ms')} if e else '')" || tail -3 $O/$name.err | cut -c1-200; }
for rep in 1 2; do
one dmtet800_1view_trained_$rep --config dmtet800 --batch 1 --graph on
NVDR_TUNING=1 NVDR_REFIT_KEEP_COLLAPSE=0 one dmtet800_1view_trained_dp_refit_$rep --config dmtet800 --batch 1 --graph on
one dmtet800_1view_trained_schedule_$rep --config dmtet800 --batch 1 --graph on --exchange-world1
NVDR_TUNING=1 NVDR_REFIT_KEEP_COLLAPSE=0 one dmtet800_1view_trained_schedule_dp_refit_$rep --config dmtet800 --batch 1 --graph on --exchange-world1
done
one dmtet800_8views_trained --config dmtet800 --steps 30 --warmup 6
NVDR_TUNING=1 NVDR_REFIT_KEEP_COLLAPSE=0 one dmtet800_8views_trained_dp_refit --config dmtet800 --steps 30 --warmup 6
one dmtet64_800_1view_trained --config dmtet64_800 --batch 1 --graph on --steps 40 --warmup 10
NVDR_TUNING=1 NVDR_REFIT_KEEP_COLLAPSE=0 one dmtet64_800_1view_trained_dp_refit --config dmtet64_800 --batch 1 --graph on --steps 40 --warmup 10
