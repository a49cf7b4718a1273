#!/bin/bash
# round 5, GPU session 31: the live-ray list as 16 segments behind their own claim counters (generation kernel) + a chunk dealer that walks the segments (traversal kernel):
# visibility / env-shade / composition parity, then A/B against commit 6e5dc40's kernels (`prev`)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s31; O=$R/gpurun_out/r5s31
timeout 1200 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -v Warning | tail -6 | tee $O/pytest.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
