#!/bin/bash
# round-2 GPU session P: is somebody else on this GPU / are our buffers in VRAM when the traversal is slow?  (sysfs + rocm-smi snapshots
# around large-mesh runs that showed the slow state in sessions K, L, M, O)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
el "== idle snapshot"
python tools/gpu_tenancy.py 2>&1 | cut -c1-600 | tee $O/r02p_tenancy.txt
timeout 20 rocm-smi --showuse --showmemuse --showmeminfo vram gtt --showpids --showclocks 2>&1 | grep -v "^$\|====" | head -40 | cut -c1-200 | tee -a $O/r02p_tenancy.txt
nproc; cat /proc/loadavg
el "== 684k triangles 8 views, single context, bursts"
for rep in 1 2 3; do PROBE_SUBDIV=3 PROBE_RES=800 PROBE_VIEWS=8 PROBE_BURST=3 PROBE_BURSTS=4 PROBE_GAPS=0,0.2 PROBE_SLOW_MS=40 timeout 120 python tools/mode_idle_probe.py 2>&1 | grep "^burst\|slow burst\|^tenancy\|Error" | cut -c1-420; echo; done | tee $O/r02p_idle_684k.txt
el "== 684k triangles 8 views, three contexts in one process (the set-up that was slow in sessions L and O)"
PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 AB_ONLY=xcdpart,lazypeek timeout 200 python tools/ab_inproc.py 2 2>&1 | grep -v amdgpu.ids | tail -7 | cut -c1-420 | tee $O/r02p_ab_684k.txt
timeout 20 rocm-smi --showuse --showmeminfo vram gtt --showpids 2>&1 | grep -v "^$\|====" | head -20 | cut -c1-200
el done
