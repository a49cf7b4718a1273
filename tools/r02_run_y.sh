#!/bin/bash
# round-2 GPU session Y: two rays per lane, software-pipelined (NVDR_TRACE_DUAL, 4 waves per SIMD): parity on the variant build, then
# interleaved in-process A/B against the current kernel.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
B=nvdiffrecmc_amd/csrc/build
cp $B/libnvdr_hip.so /tmp/lib.orig
cp $B/variants/libnvdr_hip.so.dual4 $B/libnvdr_hip.so
timeout 60 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py -x -q -m gpu -p no:cacheprovider -k "not large_mesh and not predicate" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300 | tee $O/r02y_tests.txt
cp /tmp/lib.orig $B/libnvdr_hip.so
PROBE_VIEWS=8 timeout 60 python tools/ab_inproc.py 5 2>&1 | grep -v amdgpu.ids | grep -A4 "env-shade stage" | cut -c1-200 | tee $O/r02y_ab_bob8.txt
PROBE_VIEWS=1 timeout 40 python tools/ab_inproc.py 5 2>&1 | grep -v amdgpu.ids | grep -A4 "env-shade stage" | cut -c1-200 | tee $O/r02y_ab_bob1.txt
