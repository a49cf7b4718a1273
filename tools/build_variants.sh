#!/bin/bash
# usage: build_variants.sh tag1:"-DX=1 -DY=2" tag2:"..."  -> nvdiffrecmc_amd/csrc/build/variants/libnvdr_hip.so.<tag>
# (A/B kernels must be timed inside ONE gpurun call: boxes differ by ~20 %)
set -e
R=/root/repo; B=$R/nvdiffrecmc_amd/csrc/build; mkdir -p $B/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I$R/include -I$R/nvdiffrecmc_amd/csrc -Wno-unused-result -Wno-unused-value"
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c ${SRC:-$R/nvdiffrecmc_amd/csrc/env_shade.hip} -o $B/variants/env_shade_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/core.o $B/bvh.o $B/variants/env_shade_$tag.o $B/denoise.o $B/renderutils.o $B/light.o $B/gbuffer.o $B/optim.o -ldl -o $B/variants/libnvdr_hip.so.$tag && rm $B/variants/env_shade_$tag.o && echo built $tag ) &
done
wait
