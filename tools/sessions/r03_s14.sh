#!/bin/bash
# round 3, GPU session 14: gather work split by launch size, streaming loads, finer chunks of small traversal launches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== adam + light parity"; timeout 600 python -m pytest tests -q -m gpu -x -k "adam or light or sparse or env_shade or chunk" 2>&1 | tail -3
for v in 1 2 4 8; do
echo "== A/B (in process) $v views"; AB_LG=1 AB_R2=0 PROBE_VIEWS=$v timeout 900 python tools/ab_inproc.py 4 2>&1 | grep -A6 "gen   "
done
