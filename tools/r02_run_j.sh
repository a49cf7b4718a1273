#!/bin/bash
# round-2 GPU session J: the three identical copies of the traversal kernel in fresh processes (large mesh, then bob)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 200 python -m pytest tests/test_gpu_bvh.py -x -q -m gpu -k "bit_exact or degenerate or overflow" -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -2
for rep in $(seq 1 ${REPS:-20}); do PROBE_SUBDIV=3 PROBE_RES=800 timeout 200 python tools/mode_copy_probe.py 2>&1 | grep "^traversal"; done | tee $O/r02j_copies_684k.txt
for rep in $(seq 1 ${REPS2:-12}); do timeout 100 python tools/mode_copy_probe.py 2>&1 | grep "^traversal"; done | tee $O/r02j_copies_bob.txt
