#!/bin/bash
# round 3, GPU session 27: LLVM AMDGPU scheduling strategies for env_shade.hip (all its kernels), in-process A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== A/B (in process) 8 views"; AB_R2=0 PROBE_VIEWS=8 timeout 500 python tools/ab_inproc.py 3 2>&1 | grep -A8 "gen   "
