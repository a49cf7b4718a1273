#!/bin/bash
# round 3, GPU session 23: internal children of an eight-wide node ordered by surface area (smaller first)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in 8; do
echo "== A/B (in process) $v views"; AB_R2=0 PROBE_VIEWS=$v timeout 900 python tools/ab_inproc.py 5 2>&1 | grep -A3 "gen   "
done
echo "== large mesh"; AB_R2=0 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 timeout 900 python tools/ab_inproc.py 3 2>&1 | grep -A3 "gen   "
