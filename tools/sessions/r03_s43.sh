#!/bin/bash
# round 3, GPU session 43: queue shading kernels at 256 spp (spot, 4 views); grids at 2 and 4 views of bob
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
V="q0:NVDR_SHADE_QUEUE=0;NVDR_PBLOCKS=10,6,6|q3_5_3:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,5,3|q3_15_12:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,15,12|q3_30_18:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,30,18"
AB_ENV="$V" AB_R2=0 PROBE_MESH=spot PROBE_N=16 PROBE_VIEWS=4 timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A7 "env-shade stage times"
V="q3_10_3:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,10,3|q3_10_6:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,10,6|q3_15_12:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,15,12|q3_15_9:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,15,9"
for v in 2 4; do AB_ENV="$V" AB_R2=0 PROBE_VIEWS=$v timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A7 "env-shade stage times"; done
