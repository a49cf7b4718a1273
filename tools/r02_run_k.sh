#!/bin/bash
# round-2 GPU session K (8 minutes): (1) the new tests, (2) instance selection of the traversal kernel in fresh processes (large mesh,
# then bob), (3) per-XCD chunk dealing A/B, (4) compile-time variants of the node step A/B.  Text outputs only.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
el "== new tests"
timeout 150 python -m pytest tests/test_gpu_bvh.py -x -q -m gpu -p no:cacheprovider -k "instance or bit_exact or degenerate or overflow" 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
el "== instance selection, 684k triangles, 800^2, one view, fresh processes"
for rep in $(seq 1 ${REPS:-10}); do PROBE_SUBDIV=3 PROBE_RES=800 timeout 120 python tools/mode_select_probe.py 2>&1 | grep "^select\|\[nvdr\]\|Error" | cut -c1-400; done | tee $O/r02k_select_684k.txt
el "== instance selection, bob, 512^2, one view, fresh processes"
for rep in $(seq 1 ${REPS2:-8}); do timeout 60 python tools/mode_select_probe.py 2>&1 | grep "^select\|\[nvdr\]\|Error" | cut -c1-400; done | tee $O/r02k_select_bob.txt
el "== per-XCD chunk dealing: bob 8 views, then 684k 8 views (fwd | bwd stage times)"
for x in 0 1 0 1; do NVDR_TRACE_XCD=$x PROBE_VIEWS=8 timeout 100 python tools/stage_probe.py 8,6,6 2>&1 | grep "workload\|fwd gen\|counting build" | cut -c1-330; done | tee $O/r02k_xcd_bob.txt
for x in 0 1; do NVDR_TRACE_XCD=$x PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 PROBE_REBUILD=1 timeout 150 python tools/stage_probe.py 8,6,6 2>&1 | grep "workload\|fwd gen\|counting build" | cut -c1-330; done | tee $O/r02k_xcd_684k.txt
el "== node-step variants, bob 8 views"
AB_WITH_CURRENT=1 PROBE_VIEWS=8 PROBE_CFGS=8,6,6 bash tools/ab_run.sh 2>&1 | cut -c1-330 | tee $O/r02k_variants.txt
el done
