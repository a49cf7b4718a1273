#!/bin/bash
# round 6, GPU session 6: the hang of the trained one-view schedule with a refit policy (where does it stop, does it need RCCL / the probe round / the graphs);
# A/B of the split walks and of the doubled node fetches (session 5's tables were eaten by a grep), + the quad-cooperative node fetch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s6; O=$R/gpurun_out/r6s6
run() { echo "== $*"; env "$@" timeout 150 python tools/hang_probe.py 2>&1 | grep -v "Warning\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp" | tail -12 | cut -c1-300; }
run PROBE_K=8 PROBE_SUBDIV=1 PROBE_RES=256 PROBE_STEPS=300
run PROBE_K=8 PROBE_SUBDIV=1 PROBE_RES=256 PROBE_STEPS=300 PROBE_EXCHANGE=dense
run PROBE_K=8 PROBE_SUBDIV=1 PROBE_RES=256 PROBE_STEPS=300 PROBE_RCCL=0
run PROBE_K=8 PROBE_SUBDIV=1 PROBE_RES=256 PROBE_STEPS=300 PROBE_GRAPH=0
run PROBE_K=8 PROBE_STEPS=300
bash tools/build_variants.sh nosplit:"-DNVDR_TRACE_SPLIT=0" split16:"-DNVDR_TRACE_SPLIT_FREE=16" split32e2:"-DNVDR_TRACE_SPLIT_FREE=32 -DNVDR_TRACE_SPLIT_EVERY=2" dup2:"-DNVDR_TRACE_DUP_FETCH=2" dup4:"-DNVDR_TRACE_DUP_FETCH=4" coop:"-DNVDR_TRACE_COOP_FETCH=1" 2>&1 | tail -6
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -12 | tee $O/ab_bob1.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -12 | tee $O/ab_bob8.txt
PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3 AB_ONLY=nosplit,dup4,coop timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -8 | tee $O/ab_dmtet800_1.txt
PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 AB_ONLY=nosplit,dup4,coop timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -8 | tee $O/ab_dmtet800_8.txt
timeout 600 python -c "
import sys, os
sys.path.insert(0, os.getcwd())
from nvdiffrecmc_amd import _build
_build.LIB = os.path.join(_build.BUILD, 'variants', 'libnvdr_hip.so.coop')
import pytest
sys.exit(pytest.main(['tests/test_gpu_bvh.py', '-q', '-m', 'gpu', '-x', '-k', 'visibility or invariants or 684k']))
" 2>&1 | tail -3
