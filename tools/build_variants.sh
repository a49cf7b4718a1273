#!/bin/bash
# usage: [SRC=bvh.hip] build_variants.sh tag1:"-DX=1 -DY=2" tag2:"..."  -> nvdiffrecmc_amd/csrc/build/variants/libnvdr_hip.so.<tag>
# One translation unit (default env_shade.hip) is recompiled with the extra flags and linked with the current objects of the others.
# (A/B kernels must be timed inside ONE gpurun call: boxes differ by ~20 %; tools/ab_inproc.py loads the variants side by side)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; B=$R/nvdiffrecmc_amd/csrc/build; mkdir -p $B/variants
SRC=${SRC:-env_shade.hip}; STEM=${SRC%.hip}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$R/include -I$R/nvdiffrecmc_amd/csrc -Wno-unused-result -Wno-unused-value"
OTHERS=$(ls $B/*.o | grep -v "/$STEM.o")
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c $R/nvdiffrecmc_amd/csrc/$SRC -o $B/variants/${STEM}_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $B/variants/${STEM}_$tag.o -ldl -o $B/variants/libnvdr_hip.so.$tag && rm $B/variants/${STEM}_$tag.o && echo built $tag ) &
done
wait
