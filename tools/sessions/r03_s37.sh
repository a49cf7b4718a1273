#!/bin/bash
# round 3, GPU session 37: shading kernels with the light samples queued across pixels (NVDR_SHADE_QUEUE=1) -- equality test, the env-shade
# and full-size suites under the switch, and the A/B of the stages
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "queue_shading" 2>&1 | tail -15
echo "== suites under NVDR_SHADE_QUEUE=1"
NVDR_SHADE_QUEUE=1 timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -8
echo "== A/B"
AB_ONLY=none AB_ENV="q1:NVDR_SHADE_QUEUE=1" AB_R2=0 PROBE_VIEWS=8 timeout 300 python tools/ab_inproc.py 4 2>&1 | tail -25
