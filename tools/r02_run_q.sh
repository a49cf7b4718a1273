#!/bin/bash
# round-2 GPU session Q: is the slow state of the traversal bracket a HOST-side effect (CPU quota throttling / oversubscribed threads)?
# 684k triangles, 8 views, fwd+bwd passes (the set-up that was slow in session K), default thread pools vs one thread.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python tools/gpu_tenancy.py 2>&1 | tail -1 | cut -c1-400
for rep in 1 2; do
  el "== default thread pools"
  PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 PROBE_REBUILD=1 timeout 150 python tools/stage_probe.py 8,6,6 2>&1 | grep "fwd gen\|host side\|counting build" | cut -c1-700
  el "== OMP_NUM_THREADS=1"
  OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 PROBE_REBUILD=1 timeout 150 python tools/stage_probe.py 8,6,6 2>&1 | grep "fwd gen\|host side\|counting build" | cut -c1-700
done | tee $O/r02q_host.txt
el done
