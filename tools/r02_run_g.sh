#!/bin/bash
# round-2 GPU session G: slow-mode survey per stream (12 fresh processes, bob) and the large mesh with / without a BVH rebuild per launch
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
echo "== streams, bob one view, 12 fresh processes"
for i in $(seq 1 12); do timeout 100 python tools/mode_stream_probe.py 2>&1 | grep "traversal ms"; done | tee $O/r02g_streams.txt
echo "== 684k-triangle mesh, 8 views: stage probe without / with a BVH rebuild before every launch, 2 fresh processes each"
for rep in 1 2; do
  PROBE_SUBDIV=3 PROBE_RES=800 PROBE_VIEWS=8 timeout 200 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen" | sed 's/^/no-rebuild /'
  PROBE_REBUILD=1 PROBE_SUBDIV=3 PROBE_RES=800 PROBE_VIEWS=8 timeout 200 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen" | sed 's/^/rebuild    /'
done | tee $O/r02g_rebuild.txt
echo "== 684k-triangle mesh, one view, streams, 3 fresh processes"
for i in 1 2 3; do PROBE_SUBDIV=3 PROBE_RES=800 timeout 200 python tools/mode_stream_probe.py 2>&1 | grep "traversal ms"; done | tee -a $O/r02g_streams.txt
