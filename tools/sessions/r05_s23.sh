#!/bin/bash
# round 5, GPU session 23: the integer divisions left in the per-sample / per-ray paths (light texel -> address, stream slot -> pixel): parity + A/B against commit 9e881cb (`prev`)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s23; O=$R/gpurun_out/r5s23
timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_bvh.py -q -x 2>&1 | grep -v Warning | tail -5 | tee $O/pytest.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
