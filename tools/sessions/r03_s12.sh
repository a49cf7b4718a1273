#!/bin/bash
# round 3, GPU session 12: where does the backward shading kernel's extra time come from (variants e1-e3), Adam test, counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== adam"; timeout 300 python -m pytest tests/test_gpu_renderutils.py -q -m gpu -x -k "adam" 2>&1 | tail -30
echo "== A/B (in process) 8 views"; AB_R2=0 PROBE_VIEWS=8 timeout 900 python tools/ab_inproc.py 4 2>&1 | grep -B1 -A6 "^  current"
echo "== A/B one view"; AB_R2=0 PROBE_VIEWS=1 timeout 900 python tools/ab_inproc.py 4 2>&1 | grep -B1 -A6 "^  current"
echo "== bench with counters"; timeout 600 python bench.py --no-cpu-baseline --no-large-mesh --steps 20 --warmup 5 --pmc-keep gpurun_out/pmc_s12 2>&1 | tail -1 > gpurun_out/r03_bench_s12.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_s12.json'))
print(d['ms_per_step'], d['median_ms_per_step'], d['value'])
print(json.dumps(d['roofline'].get('other_kernels'),indent=0))
PY
