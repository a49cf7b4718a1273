#!/bin/bash
# round 3, GPU session 24: traversal kernel with early refill (generation-tagged FIFO of triangle tests)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
B=nvdiffrecmc_amd/csrc/build
for v in 8 1; do
echo "== A/B (in process) $v views"; AB_R2=0 PROBE_VIEWS=$v timeout 900 python tools/ab_inproc.py 4 2>&1 | grep -A5 "gen   "
done
echo "== large mesh"; AB_R2=0 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 timeout 900 python tools/ab_inproc.py 3 2>&1 | grep -A5 "gen   "
echo "== parity of the early8 variant"
cp $B/libnvdr_hip.so /tmp/lib.orig; cp $B/variants/libnvdr_hip.so.early8 $B/libnvdr_hip.so
mv $B/variants /tmp/variants_aside
timeout 900 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -5
cp /tmp/lib.orig $B/libnvdr_hip.so; mv /tmp/variants_aside $B/variants
