#!/bin/bash
# round 3, GPU session 26: occupancy / grid retune of the shading and generation kernels after this round's changes; shading_frame test
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== tests"; timeout 300 python -m pytest tests/test_gpu_renderutils.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -3
echo "== A/B (in process) 8 views"; AB_ENV="b4:NVDR_PBLOCKS=8,6,4|b5:NVDR_PBLOCKS=8,6,5|b8:NVDR_PBLOCKS=8,6,8|f4:NVDR_PBLOCKS=8,4,6|f8:NVDR_PBLOCKS=8,8,6" AB_R2=0 PROBE_VIEWS=8 timeout 400 python tools/ab_inproc.py 3 2>&1 | grep -A11 "gen   "
echo "== A/B (in process) 1 view"; AB_ENV="b4:NVDR_PBLOCKS=8,6,4|b8:NVDR_PBLOCKS=8,6,8" AB_R2=0 PROBE_VIEWS=1 timeout 300 python tools/ab_inproc.py 3 2>&1 | grep -A8 "gen   "
