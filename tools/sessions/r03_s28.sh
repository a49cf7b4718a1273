#!/bin/bash
# round 3, GPU session 28: CDF inversion through guide tables (parity: bit-exact against the oracle's bisection; A/B)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== parity"; timeout 400 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_renderutils.py -q -m gpu -x 2>&1 | tail -4
for v in 8 1; do echo "== A/B $v views"; AB_R2=0 PROBE_VIEWS=$v timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A3 "gen   "; done
