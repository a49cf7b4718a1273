#!/bin/bash
# round 6, GPU session 33: backward shading (64 spp): a pixel's twelve result rows summed by 48 lanes from LDS instead of twelve 64-lane butterflies: parity, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s33; O=$R/gpurun_out/r6s33
bash tools/build_variants.sh butterfly:"-DNVDR_SQ_ROW_SUMS=0" 2>&1 | tail -1
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py -q -m gpu -x 2>&1 | grep -v Warning | tail -3 | tee $O/tests.txt
ab() { out=$1; shift; env "$@" timeout 900 python tools/ab_inproc.py ${ROUNDS:-5} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -5 | tee $O/$out.txt; }
ab ab_bob8 PROBE_VIEWS=8
ab ab_bob1 PROBE_VIEWS=1
ab ab_684k_1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3
ROUNDS=3 ab ab_684k_8 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3
