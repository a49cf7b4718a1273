#!/bin/bash
# round-2 GPU session R: is the stall of the traversal bracket caused by the tiny kernel (or memset) that runs right before the persistent
# kernel?  NVDR_DEBUG=64 drops the 64-thread reset kernel (stages 1 and 3 leave the counters zeroed).  684k triangles, 8 views, fwd+bwd.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 100 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-200
NVDR_DEBUG=64 timeout 100 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|NVDR_DEBUG" | tail -2 | cut -c1-200
for rep in 1 2 3; do
  for dbg in 0 64; do
    el "== NVDR_DEBUG=$dbg"
    NVDR_DEBUG=$dbg PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 PROBE_REBUILD=1 timeout 150 python tools/stage_probe.py 8,6,6 2>&1 | grep "fwd gen\|host side\|counting build\|wave begin\|wave end" | cut -c1-330
  done
done | tee $O/r02r_notiny.txt
el done
