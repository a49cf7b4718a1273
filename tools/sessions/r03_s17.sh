#!/bin/bash
# round 3, GPU session 17: SAH-optimal collapse (DP in the fit kernel), generation kernel grid, BSDF rays listed at push time
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== gpu tests"; timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -6
for v in 8 1; do
echo "== A/B (in process) $v views"; AB_ENV="greedy:NVDR_OCT_DP=0|cl30:NVDR_OCT_CLEAF=0.3|cl70:NVDR_OCT_CLEAF=0.7|gen4:NVDR_PBLOCKS=4,6,6|gen6:NVDR_PBLOCKS=6,6,6" AB_R2=0 PROBE_VIEWS=$v timeout 900 python tools/ab_inproc.py 4 2>&1 | grep -A8 "gen   "
done
echo "== large mesh"; AB_ENV="greedy:NVDR_OCT_DP=0|cl30:NVDR_OCT_CLEAF=0.3|cl70:NVDR_OCT_CLEAF=0.7" AB_R2=0 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 timeout 900 python tools/ab_inproc.py 3 2>&1 | grep -A6 "gen   "
python tools/stage_probe.py 6,6,4 2>&1 | tail -4
