#!/bin/bash
# round 3, GPU session 30: generation grid after the guide tables; tile height of the pair filter
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== gen grid"; AB_ONLY=none AB_ENV="g6:NVDR_PBLOCKS=6,6,6|g10:NVDR_PBLOCKS=10,6,6|g12:NVDR_PBLOCKS=12,6,6" AB_R2=0 PROBE_VIEWS=8 timeout 300 python tools/ab_inproc.py 4 2>&1 | grep -A5 "gen   "
echo "== pair filter tiles"; PROBE_VIEWS=8 timeout 300 python tools/dn_probe.py 5 2>&1 | tail -5
