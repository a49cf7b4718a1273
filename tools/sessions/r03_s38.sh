#!/bin/bash
# round 3, GPU session 38: A/B of the stages with the queue shading kernels, 8 views and 1 view
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in 8 1; do
AB_ENV="q1:NVDR_SHADE_QUEUE=1" AB_R2=0 PROBE_VIEWS=$v timeout 300 python tools/ab_inproc.py 5 2>&1 | grep -A4 "env-shade stage times"
done
