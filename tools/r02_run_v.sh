#!/bin/bash
# round-2 GPU session V: CDF bisection with 2 / 3 steps per round trip in the sample-generation kernel (bit-identical by construction):
# interleaved in-process A/B (bob 8 views, spot 256 spp is the same kernel), then the env-shade parity tests on the cdf3 build.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
PROBE_VIEWS=8 timeout 120 python tools/ab_inproc.py 6 2>&1 | grep -v amdgpu.ids | grep -A6 "env-shade stage" | cut -c1-200 | tee $O/r02v_ab_cdf.txt
B=nvdiffrecmc_amd/csrc/build
cp $B/libnvdr_hip.so /tmp/lib.orig
for v in cdf3 cdf2; do
  cp $B/variants/libnvdr_hip.so.$v $B/libnvdr_hip.so
  echo "== parity tests on $v"
  timeout 150 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
done | tee $O/r02v_tests.txt
cp /tmp/lib.orig $B/libnvdr_hip.so
