#!/bin/bash
# round-2 GPU session I: per-XCD wave times of the large-mesh traversal launch in fresh processes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
for rep in $(seq 1 ${REPS:-24}); do PROBE_SUBDIV=3 PROBE_RES=800 timeout 200 python tools/mode_xcd_probe.py 2>&1 | grep "^trace"; done | tee $O/r02i_xcd.txt
