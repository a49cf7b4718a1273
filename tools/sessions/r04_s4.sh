#!/bin/bash
# round 4, GPU session 4: the geometry / material gradient route (csrc/mesh.hip) + the iteration with the reference's parameter set
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s4; O=$R/gpurun_out/r4s4
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_renderutils.py tests/test_gpu_gbuffer.py -q 2>&1 | tail -60
