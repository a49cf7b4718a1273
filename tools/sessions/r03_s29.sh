#!/bin/bash
# round 3, GPU session 29: starting dependent-load chains early in the per-pixel kernels (next pixel's index; texel -> radiance)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in 8 1; do echo "== A/B $v views"; AB_R2=0 PROBE_VIEWS=$v timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A5 "gen   "; done
