#!/bin/bash
# round-2 GPU session T: do the 0.3-0.4 s stalls of the traversal kernel need the chunk-claim atomics / the HBM stack spill / the full
# residency?  684k triangles, 8 views, variants interleaved in one process, one measured pass per round (per-pass times printed).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
rocminfo 2>/dev/null | grep -i -m4 "xnack\|Marketing" | cut -c1-120; cat /sys/module/amdgpu/parameters/noretry 2>/dev/null; echo "HSA_XNACK=$HSA_XNACK"
PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 AB_ITERS=2 timeout 240 python tools/ab_inproc.py 10 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-400 | tee $O/r02t_ab_684k.txt
