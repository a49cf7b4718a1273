#!/bin/bash
# round 6, GPU session 27: the fp64 gather again, now that the shading kernel's record placement no longer costs per band; then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s27; O=$R/gpurun_out/r6s27
E="f64:NVDR_LG_F64=1|f64perband:NVDR_LG_F64=1;NVDR_LG_MODE=1"
ab() { out=$1; shift; env "$@" AB_ENV="$E" timeout 900 python tools/ab_inproc.py ${ROUNDS:-5} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -6 | tee $O/$out.txt; }
ab ab_bob1 PROBE_VIEWS=1
ab ab_bob8 PROBE_VIEWS=8
ab ab_684k_1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3
ROUNDS=3 ab ab_684k_8 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3
echo "== gpu tests"; timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
