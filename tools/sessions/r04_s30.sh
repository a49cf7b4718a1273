#!/bin/bash
# round 4, GPU session 30: children blocks of the eight-wide tree on 128-byte lines: structure tests + in-process A/B against the unaligned layout
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s30; O=$R/gpurun_out/r4s30
timeout 900 python -m pytest tests/test_gpu_bvh.py -q 2>&1 | tail -3
for cfg in "bob 0 512 8" "bob 3 800 8" "bob 2 512 8"; do set -- $cfg
echo "== $1 subdiv $2 res $3 views $4"
AB_R2=0 PROBE_MESH=$1 PROBE_SUBDIV=$2 PROBE_RES=$3 PROBE_VIEWS=$4 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -A4 "env-shade stage times" | cut -c1-220
done
