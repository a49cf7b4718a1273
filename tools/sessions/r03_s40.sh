#!/bin/bash
# round 3, GPU session 40: grids of the queue shading kernels (workgroups per CU: forward, backward)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
V="q_15_12:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,15,12|q_20_15:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,20,15|q_30_18:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,30,18|q_40_24:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,40,24|q_60_36:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,60,36"
AB_ENV="$V" AB_R2=0 PROBE_VIEWS=8 timeout 300 python tools/ab_inproc.py 5 2>&1 | grep -A8 "env-shade stage times"
V="q_5_3:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,5,3|q_10_6:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,10,6|q_15_12:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,15,12|q_20_15:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,20,15"
AB_ENV="$V" AB_R2=0 PROBE_VIEWS=1 timeout 300 python tools/ab_inproc.py 5 2>&1 | grep -A7 "env-shade stage times"
