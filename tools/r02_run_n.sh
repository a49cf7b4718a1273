#!/bin/bash
# round-2 GPU session N: does the slow state of the traversal kernel re-roll across idle gaps?  (bursts + gaps inside one process)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
el "== bob, one view, bursts of 30 launches"
for rep in 1 2 3; do timeout 60 python tools/mode_idle_probe.py 2>&1 | grep "^burst\|^first\|Error" | cut -c1-260; echo; done | tee $O/r02n_idle_bob.txt
el "== 684k triangles, 800^2, 8 views, bursts of 3 launches"
for rep in 1 2; do PROBE_SUBDIV=3 PROBE_RES=800 PROBE_VIEWS=8 PROBE_BURST=3 PROBE_BURSTS=12 timeout 120 python tools/mode_idle_probe.py 2>&1 | grep "^burst\|^first\|Error" | cut -c1-260; echo; done | tee $O/r02n_idle_684k.txt
el done
