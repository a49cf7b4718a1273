#!/bin/bash
# round 3, GPU session 25: early-refill traversal variant after the permute fix: parity first (short timeouts), then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
B=nvdiffrecmc_amd/csrc/build
echo "== parity of the early8 variant"
cp $B/libnvdr_hip.so /tmp/lib.orig; cp $B/variants/libnvdr_hip.so.early8 $B/libnvdr_hip.so
mv $B/variants /tmp/variants_aside
timeout 200 python -m pytest tests/test_gpu_bvh.py -q -m gpu -x 2>&1 | tail -4
ok=$?
timeout 150 python -m pytest tests/test_gpu_env_shade.py -q -m gpu -x 2>&1 | tail -3
cp /tmp/lib.orig $B/libnvdr_hip.so; mv /tmp/variants_aside $B/variants
echo "== A/B (in process) 8 views"; AB_R2=0 PROBE_VIEWS=8 timeout 240 python tools/ab_inproc.py 3 2>&1 | grep -A5 "gen   "
echo "== A/B (in process) 1 view"; AB_R2=0 PROBE_VIEWS=1 timeout 200 python tools/ab_inproc.py 3 2>&1 | grep -A5 "gen   "
