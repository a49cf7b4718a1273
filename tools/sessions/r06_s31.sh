#!/bin/bash
# round 6, GPU session 31: the per-band rule of the gather raised to cover the 8-view benchmark launch (fp64 + per band): A/B against the previous rule (all bands, fp32), then the measurement set
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s31; O=$R/gpurun_out/r6s31
E="allbands_f32:NVDR_LG_MODE=0;NVDR_LG_F64=0"
ab() { out=$1; shift; env "$@" AB_ENV="$E" timeout 900 python tools/ab_inproc.py ${ROUNDS:-6} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -5 | tee $O/$out.txt; }
ab ab_bob8 PROBE_VIEWS=8
ab ab_bob8_b PROBE_VIEWS=8
ROUNDS=3 ab ab_spot8 PROBE_MESH=spot PROBE_VIEWS=8
bash tools/r06_final.sh
