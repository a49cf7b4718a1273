#!/bin/bash
# round 6, GPU session 19: the treetop of the eight-wide tree in LDS (bvh.h NVDR_TRACE_TOP): parity, then in-process A/B over the table's size
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s19; O=$R/gpurun_out/r6s19
bash tools/build_variants.sh notop:"-DNVDR_TRACE_TOP_MAX=0" 2>&1 | tail -2
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py -q -m gpu -x -s -k "treetop or visibility or invariants or 684k or vs_oracle or degenerate or overflow" 2>&1 | grep -v Warning | grep "treetop of\|passed\|failed\|Error\|assert" | tee $O/tests.txt
E="top0:NVDR_TRACE_TOP_NODES=0|top9:NVDR_TRACE_TOP_NODES=9|top32:NVDR_TRACE_TOP_NODES=32|top128:NVDR_TRACE_TOP_NODES=128|top256:NVDR_TRACE_TOP_NODES=256"
ab() { out=$1; shift; env "$@" AB_ENV="$E" timeout 900 python tools/ab_inproc.py ${ROUNDS:-4} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -12 | tee $O/$out.txt; }
ab ab_bob8 PROBE_VIEWS=8
ab ab_bob1 PROBE_VIEWS=1
ab ab_684k_1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3
ROUNDS=3 ab ab_684k_8 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3
ab ab_dmtet64_1 PROBE_MESH=dmtet64_mid PROBE_VIEWS=1 PROBE_RES=800
