#!/bin/bash
# round 5, GPU session 20: csrc/ieee_arith.h -- the device self-test again (negative denormals moved out of the square root's domain) and the parity suites of the shading kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s20; O=$R/gpurun_out/r5s20
timeout 900 python -m pytest tests/test_gpu_arith.py -q -s 2>&1 | grep -v Warning | tail -15 | tee $O/pytest_arith.txt
timeout 1200 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_renderutils.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py tests/test_gpu_denoiser.py -q -x 2>&1 | grep -v Warning | tail -15 | tee $O/pytest.txt
