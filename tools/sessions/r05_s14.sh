#!/bin/bash
# round 5, GPU session 14: + the incoming gradients packed per compacted pixel (no strided view left in the backward shading kernels): parity suites + in-process A/B against the 64-bit build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s14; O=$R/gpurun_out/r5s14
timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py tests/test_gpu_gbuffer.py -q 2>&1 | grep -E "^E |passed|failed" | head -20 | cut -c1-300 | tee $O/pytest.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
PROBE_VIEWS=4 PROBE_MESH=spot PROBE_N=16 AB_ITERS=3 timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v Warning | tee $O/ab_spot4.txt
