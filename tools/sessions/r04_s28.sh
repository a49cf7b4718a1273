#!/bin/bash
# round 4, GPU session 28: direction-octant window sort of the live-ray list: parity (env-shade / fullsize / bvh suites) and in-process A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s28; O=$R/gpurun_out/r4s28
timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_bvh.py -q 2>&1 | tail -3
for cfg in "bob 0 512 8" "bob 0 512 1" "bob 3 800 8" "bob 3 800 1"; do set -- $cfg
echo "== $1 subdiv $2 res $3 views $4"
AB_R2=0 AB_ENV="nosort:NVDR_LIVE_SORT=0" PROBE_MESH=$1 PROBE_SUBDIV=$2 PROBE_RES=$3 PROBE_VIEWS=$4 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -A4 "env-shade stage times" | cut -c1-220
done
