#!/bin/bash
# round 6, GPU session 20: (a) the treetop build as a kernel of its own, off by default: its test; (b) A/B of a one-dword TOUCH of the node a lane visits next, issued at the end of its node step
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s20; O=$R/gpurun_out/r6s20
bash tools/build_variants.sh touch:"-DNVDR_TRACE_TOUCH=1" 2>&1 | tail -2
echo "== tests"; timeout 1200 python -m pytest tests/test_gpu_bvh.py -q -m gpu -x -s -k "treetop or split_walk" 2>&1 | grep -v Warning | grep "treetop of\|passed\|failed\|Error\|assert" | tee $O/tests.txt
timeout 900 python -c "
import sys, os
sys.path.insert(0, os.getcwd())
from nvdiffrecmc_amd import _build
_build.LIB = os.path.join(_build.BUILD, 'variants', 'libnvdr_hip.so.touch')
import pytest
sys.exit(pytest.main(['tests/test_gpu_bvh.py', 'tests/test_gpu_env_shade.py', '-q', '-m', 'gpu', '-x', '-k', 'visibility or 684k or vs_oracle or degenerate']))
" 2>&1 | tail -3
E="top64:NVDR_TRACE_TOP_NODES=64"
ab() { out=$1; shift; env "$@" AB_ENV="$E" timeout 900 python tools/ab_inproc.py ${ROUNDS:-5} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -6 | tee $O/$out.txt; }
ab ab_bob8 PROBE_VIEWS=8
ab ab_bob1 PROBE_VIEWS=1
ab ab_684k_1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3
ROUNDS=3 ab ab_684k_8 PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3
ab ab_dmtet64_1 PROBE_MESH=dmtet64_mid PROBE_VIEWS=1 PROBE_RES=800
