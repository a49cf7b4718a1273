#!/bin/bash
# round 5, GPU session 19: division / square root without the compiler's range-scaling steps (csrc/ieee_arith.h): the device self-test over every
# float, the parity suites of the shading kernels, and the A/B against the plain operators (variant `plain` = -DNVDR_PLAIN_ARITH=1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s19; O=$R/gpurun_out/r5s19
timeout 900 python -m pytest tests/test_gpu_arith.py tests/test_gpu_env_shade.py tests/test_gpu_renderutils.py tests/test_gpu_fullsize.py -q -x -s 2>&1 | grep -v Warning | tail -15 | tee $O/pytest.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
