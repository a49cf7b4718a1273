#!/bin/bash
# round 3, GPU session 39: queue shading kernels with the set-up in registers for the in-place pass; grids that are multiples of what is resident
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "queue_shading" 2>&1 | tail -3
V="q3:NVDR_SHADE_QUEUE=3|q3_5_6:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,5,6|q3_10_9:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,10,9|q3_5_3:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,5,3|q3_15_12:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,15,12"
AB_ENV="$V" AB_R2=0 PROBE_VIEWS=8 timeout 300 python tools/ab_inproc.py 5 2>&1 | grep -A8 "env-shade stage times"
V="q3:NVDR_SHADE_QUEUE=3|q3_5_6:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,5,6|q3_5_3:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,5,3|q3_10_9:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=4,10,9"
AB_ENV="$V" AB_R2=0 PROBE_VIEWS=1 timeout 300 python tools/ab_inproc.py 5 2>&1 | grep -A7 "env-shade stage times"
