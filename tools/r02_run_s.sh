#!/bin/bash
# round-2 GPU session S: where does the time of a slow traversal bracket go?  (1) event bracket of the counting launch next to its
# in-kernel clocks, (2) rocprofv3 kernel trace of the same probe: the kernel's own duration and the idle gaps around it.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
for rep in 1 2; do
  el "== plain"
  PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 PROBE_REBUILD=1 timeout 150 python tools/stage_probe.py 8,6,6 2>&1 | grep "fwd gen\|counting" | cut -c1-330
  el "== under rocprofv3 --kernel-trace"
  cd /tmp; rm -rf /tmp/kt
  PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 PROBE_REBUILD=1 timeout 200 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/tools/stage_probe.py 8,6,6 2>&1 | grep "fwd gen\|counting" | cut -c1-330
  DB=$(find /tmp/kt -name "*_results.db" | head -1)
  for k in "env_trace_kernel<false>" "env_gen_kernel" "env_shade_kernel<true>"; do timeout 60 python $R/tools/rocpd_timeline.py $DB "$k" | cut -c1-600; done
  cd $R
done | tee $O/r02s_timeline.txt
el done
