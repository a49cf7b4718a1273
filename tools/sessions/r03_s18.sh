#!/bin/bash
# round 3, GPU session 18: light samples of the shading kernels queued across pixels (parity, A/B against in-place shading)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== gpu tests"; timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -12
for v in 8 1; do
echo "== A/B (in process) $v views"; AB_R2=0 PROBE_VIEWS=$v timeout 900 python tools/ab_inproc.py 4 2>&1 | grep -A3 "gen   "
done
