#!/bin/bash
# round 3, GPU session 7: gather (mask extraction) check + straggler survey over fresh processes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== tests (env_shade, fullsize)"; timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -4
echo "== kernel trace 8 views"; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 10 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_s7.md > /dev/null; cut -c1-60,100-175 $R/gpurun_out/r03_kernel_trace_s7.md | head -12
cd $R
export NVDR_MESH_CACHE=/tmp
echo "== one-view iteration, 14 fresh processes (graph on)"
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do timeout 200 python bench.py --no-cpu-baseline --no-pmc --no-large-mesh --batch 1 --graph on --steps 60 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r03_oneview_fresh_$i.json; python -c "
import json; j=json.loads(open('gpurun_out/r03_oneview_fresh_$i.json').read()); print($i, {k: round(j[k],3) if isinstance(j[k], float) else j[k] for k in ('ms_per_step','median_ms_per_step','max_ms_per_step','steps_over_twice_the_median','median_over_steps')})"; done
echo "== dmtet800, 5 fresh processes"
for i in 1 2 3 4 5; do timeout 300 python bench.py --config dmtet800 --no-cpu-baseline --no-pmc --steps 10 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03_dmtet800_fresh_$i.json; python -c "
import json; j=json.loads(open('gpurun_out/r03_dmtet800_fresh_$i.json').read()); print($i, {k: round(j[k],3) if isinstance(j[k], float) else j[k] for k in ('ms_per_step','median_ms_per_step','max_ms_per_step','steps_over_twice_the_median','median_over_steps')}, round(j['roofline']['kernel_ms_hip_events'],3))"; done
