#!/bin/bash
# round 3, GPU session 15: full parity suite after the exact division replacements; streaming loads; bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== adam"; timeout 300 python -m pytest tests/test_gpu_renderutils.py -q -m gpu -x -k "adam" 2>&1 | grep -E "Error|error|assert|passed|failed" | head -20
echo "== gpu tests"; timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -8
for v in 1 8; do
echo "== A/B (in process) $v views"; AB_R2=0 PROBE_VIEWS=$v timeout 900 python tools/ab_inproc.py 4 2>&1 | grep -A4 "gen   "
done
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-large-mesh --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8 views', d['ms_per_step'], d['median_ms_per_step'], d['value'])"
timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-pmc --no-large-mesh --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 view', d['hip_graph'], d['ms_per_step'], d['median_ms_per_step'])"
