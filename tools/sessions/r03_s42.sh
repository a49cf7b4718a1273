#!/bin/bash
# round 3, GPU session 42: queue shading kernels for every S > 32 (several rounds per pixel): tests, A/B at 256 spp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "queue_shading" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -4
V="q0:NVDR_SHADE_QUEUE=0|q0b:NVDR_SHADE_QUEUE=0;NVDR_PBLOCKS=10,6,6|q3_5_3:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,5,3|q3_30_18:NVDR_SHADE_QUEUE=3;NVDR_PBLOCKS=10,30,18"
AB_ENV="$V" AB_R2=0 PROBE_MESH=spot PROBE_N=16 PROBE_VIEWS=4 timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A7 "env-shade stage times"
