#!/bin/bash
# time every library variant under nvdiffrecmc_amd/csrc/build/variants/ (and the current build) in ONE gpurun call
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
B=nvdiffrecmc_amd/csrc/build
cp $B/libnvdr_hip.so /tmp/libnvdr_hip.so.orig
first=1
for f in ${AB_WITH_CURRENT:+/tmp/libnvdr_hip.so.orig} $B/variants/libnvdr_hip.so.*; do
  tag=${f##*.so.}
  cp $f $B/libnvdr_hip.so
  echo "== $tag"
  timeout 180 python tools/stage_probe.py ${PROBE_CFGS:-8,6,6} 2>&1 | grep "workload\|fwd gen\|counting\|Error\|error"
  if [ "$AB_TEST" = "all" ] || { [ -n "$AB_TEST" ] && [ $first = 1 ]; }; then timeout 600 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_bvh.py -q -m gpu -x 2>&1 | tail -2; fi
  first=0
done
cp /tmp/libnvdr_hip.so.orig $B/libnvdr_hip.so
