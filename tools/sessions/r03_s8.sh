#!/bin/bash
# round 3, GPU session 8: where does the light-gradient gather spend its time?  (elimination variants, kernel trace of each)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
B=nvdiffrecmc_amd/csrc/build
cp $B/libnvdr_hip.so /tmp/libnvdr_hip.so.orig
for tag in orig lg1 lg2 lg3; do
  if [ $tag = orig ]; then cp /tmp/libnvdr_hip.so.orig $B/libnvdr_hip.so; else cp $B/variants/libnvdr_hip.so.$tag $B/libnvdr_hip.so; fi
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
  timeout 200 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 6 --warmup 3 > /tmp/kt.log 2>&1
  echo "== $tag"; timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db /tmp/kt_$tag.md > /dev/null; grep "light_grad_band\|env_shade_kernel<true>" /tmp/kt_$tag.md | cut -c1-40,100-180
  cd $R
done
cp /tmp/libnvdr_hip.so.orig $B/libnvdr_hip.so
