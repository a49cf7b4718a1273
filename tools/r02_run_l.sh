#!/bin/bash
# round-2 GPU session L: in-process interleaved A/B of the node-step variants and of the per-XCD dealing (bob, then 684k triangles),
# then the slow-PHASE probe (does another process's memory teardown / this process's own hipFree slow the traversal down?).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
el "== in-process A/B, bob 8 views"
PROBE_VIEWS=8 timeout 150 python tools/ab_inproc.py 6 2>&1 | grep -v amdgpu.ids | tail -14 | tee $O/r02l_ab_bob.txt
el "== in-process A/B, bob 1 view"
PROBE_VIEWS=1 AB_ONLY=both,unord,leaf12,xcdpart timeout 100 python tools/ab_inproc.py 6 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/r02l_ab_bob1.txt
el "== in-process A/B, 684k triangles 800^2, 8 views"
PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 AB_ONLY=both,unord,xcdpart timeout 200 python tools/ab_inproc.py 4 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/r02l_ab_684k.txt
el "== slow-phase probe right after the previous process exited"
timeout 100 python tools/mode_time_probe.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-1500 | tee $O/r02l_phase_a.txt
el "== slow-phase probe after 8 s of quiet"
sleep 8
timeout 100 python tools/mode_time_probe.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-1500 | tee $O/r02l_phase_b.txt
el done
