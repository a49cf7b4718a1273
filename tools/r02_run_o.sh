#!/bin/bash
# round-2 GPU session O: in-process interleaved A/B of the second batch of env-shade variants (node-step pick, lazy peek, occupancies,
# leaf batch size, LDS stack depth), all five stage times; bob 8 views, bob 1 view, 684k triangles 8 views (+ per-XCD dealing).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
el "== bob 8 views"
PROBE_VIEWS=8 timeout 200 python tools/ab_inproc.py 5 2>&1 | grep -v amdgpu.ids | tail -17 | cut -c1-200 | tee $O/r02o_ab_bob8.txt
el "== bob 1 view"
PROBE_VIEWS=1 AB_ONLY=pick1,pick2,lazypeek,occ6,leaf12,leaf16,stack16 timeout 100 python tools/ab_inproc.py 6 2>&1 | grep -v amdgpu.ids | tail -11 | cut -c1-200 | tee $O/r02o_ab_bob1.txt
el "== 684k triangles 800^2, 8 views"
PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 AB_ONLY=pick1,pick2,lazypeek,occ6,xcdpart timeout 200 python tools/ab_inproc.py 3 2>&1 | grep -v amdgpu.ids | tail -10 | cut -c1-200 | tee $O/r02o_ab_684k.txt
el done
