#!/bin/bash
# round-2 GPU session M: kernel or memset?  (mode_memset_probe in fresh processes, one view and eight views of the 684k mesh), then the
# whole GPU test suite on the new default build (unordered descent, counter reset by a kernel).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp NVDR_MESH_CACHE=/tmp
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 )) s] $*"; }
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
el "== kernel or memset, 684k triangles, one view"
for rep in 1 2 3 4 5 6; do PROBE_SECONDS=4 timeout 60 python tools/mode_memset_probe.py 2>&1 | grep "^first\|Error" | cut -c1-500; done | tee $O/r02m_memset_1view.txt
el "== kernel or memset, 684k triangles, eight views"
for rep in 1 2 3; do PROBE_VIEWS=8 PROBE_SECONDS=5 timeout 90 python tools/mode_memset_probe.py 2>&1 | grep "^first\|Error" | cut -c1-500; done | tee $O/r02m_memset_8view.txt
el "== kernel or memset, bob, one view"
for rep in 1 2 3 4; do PROBE_SUBDIV=0 PROBE_RES=512 PROBE_SECONDS=2 timeout 60 python tools/mode_memset_probe.py 2>&1 | grep "^first\|Error" | cut -c1-500; done | tee $O/r02m_memset_bob.txt
el "== GPU test suite"
timeout 400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -5 | cut -c1-300 | tee $O/r02m_pytest.txt
el done
