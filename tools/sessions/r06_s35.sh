#!/bin/bash
# round 6, GPU session 35: row sums in the backward queue kernel only (the forward keeps its butterfly), quarters rotated per row against LDS bank conflicts (A/B against unrotated), then the measurement set
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s35; O=$R/gpurun_out/r6s35
bash tools/build_variants.sh norot:"-DNVDR_SQ_ROW_ROT=0" butterfly:"-DNVDR_SQ_ROW_SUMS=0" 2>&1 | tail -2
ab() { out=$1; shift; env "$@" timeout 900 python tools/ab_inproc.py ${ROUNDS:-5} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -6 | tee $O/$out.txt; }
ab ab_bob8 PROBE_VIEWS=8
ab ab_684k_1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3
rm -f nvdiffrecmc_amd/csrc/build/variants/*
bash tools/r06_final.sh
