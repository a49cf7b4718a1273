#!/bin/bash
# round 3, GPU session 13: backward shading kernel variants (no records / round-robin groups / in-place records / streaming loads)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== adam"; timeout 300 python -m pytest tests/test_gpu_renderutils.py -q -m gpu -x -k "adam" 2>&1 | tail -3
echo "== A/B (in process) 8 views"; AB_R2=0 PROBE_VIEWS=8 timeout 900 python tools/ab_inproc.py 4 2>&1 | tail -16
echo "== A/B one view"; AB_R2=0 PROBE_VIEWS=1 timeout 900 python tools/ab_inproc.py 4 2>&1 | tail -16
