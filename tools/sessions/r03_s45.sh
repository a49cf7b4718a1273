#!/bin/bash
# round 3, GPU session 45: queue kernels with ONE lane reduction per pixel (rows added per lane first): tests, 256 spp A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "queue_shading" 2>&1 | tail -3
V="q7_5_3:NVDR_SHADE_QUEUE=7;NVDR_PBLOCKS=10,5,3|q7_15_12:NVDR_SHADE_QUEUE=7;NVDR_PBLOCKS=10,15,12|q7_10_6:NVDR_SHADE_QUEUE=7;NVDR_PBLOCKS=10,10,6"
AB_ENV="$V" AB_R2=0 PROBE_MESH=spot PROBE_N=16 PROBE_VIEWS=4 timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A6 "env-shade stage times"
AB_ENV="q0:NVDR_SHADE_QUEUE=0" AB_R2=0 PROBE_VIEWS=8 timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A4 "env-shade stage times"
