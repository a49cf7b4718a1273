#!/bin/bash
# round 3, GPU session 20: live light samples queued in the generation kernel (pdfs of live samples only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== gpu tests"; timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -8
for v in 8 1; do
echo "== A/B (in process) $v views"; AB_R2=0 PROBE_VIEWS=$v timeout 900 python tools/ab_inproc.py 4 2>&1 | grep -A3 "gen   "
done
