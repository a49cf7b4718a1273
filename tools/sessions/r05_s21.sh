#!/bin/bash
# round 5, GPU session 21: the generation kernel's per-pixel set-up run one pixel per lane for a whole ring cycle (+ stratum / n as a multiplication):
# parity suites, then A/B against the previous kernel (variant `prevgen` = env_shade.hip of commit 3f94a86) and the plain operators (`plain`)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s21; O=$R/gpurun_out/r5s21
timeout 1200 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -v Warning | tail -15 | tee $O/pytest.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
PROBE_VIEWS=4 PROBE_MESH=spot PROBE_N=16 timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v Warning | tee $O/ab_spot4_256spp.txt
