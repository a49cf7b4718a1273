#!/bin/bash
# round 3, GPU session 46: back to the one-round queue kernels (S = 64 only): tests, 8-view A/B against the plain kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -3
AB_ENV="q0:NVDR_SHADE_QUEUE=0" AB_R2=0 PROBE_VIEWS=8 timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A4 "env-shade stage times"
AB_ENV="q0:NVDR_SHADE_QUEUE=0" AB_R2=0 PROBE_VIEWS=4 timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A4 "env-shade stage times"
