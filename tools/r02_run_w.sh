#!/bin/bash
# round-2 GPU session W: vector-L1 lookup micro-benchmark (tools/ubench/l1_lookup.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
cd tools/ubench && hipcc --offload-arch=gfx950 -O3 l1_lookup.hip -o l1_lookup 2>/dev/null; timeout 60 ./l1_lookup | tee $R/gpurun_out/r02w_l1_lookup.txt
