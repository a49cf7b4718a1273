#!/bin/bash
# round-2 GPU session H: does the mapping of the BVH arrays decide the per-process slow mode?  684k-triangle mesh, one view, fresh
# processes, three allocation modes interleaved.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
for rep in $(seq 1 ${REPS:-16}); do for mode in default contig align2m; do
  echo -n "$mode: "; NVDR_BVH_ALLOC=$mode PROBE_SUBDIV=3 PROBE_RES=800 timeout 200 python tools/mode_stream_probe.py 2>&1 | grep "traversal ms" | sed 's/| new-stream.*| default-again/| again/'
done; done | tee $O/r02h_alloc.txt
