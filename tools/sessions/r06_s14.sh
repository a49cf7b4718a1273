#!/bin/bash
# round 6, GPU session 14: split walks, second design (the OLDEST pending group goes to an idle lane): parity with the variant library, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s14; O=$R/gpurun_out/r6s14
bash tools/build_variants.sh split:"-DNVDR_TRACE_SPLIT=1" split16e2:"-DNVDR_TRACE_SPLIT=1 -DNVDR_TRACE_SPLIT_FREE=16 -DNVDR_TRACE_SPLIT_EVERY=2" split32e8:"-DNVDR_TRACE_SPLIT=1 -DNVDR_TRACE_SPLIT_FREE=32 -DNVDR_TRACE_SPLIT_EVERY=8" split8e1:"-DNVDR_TRACE_SPLIT=1 -DNVDR_TRACE_SPLIT_FREE=8 -DNVDR_TRACE_SPLIT_EVERY=1" 2>&1 | tail -4
timeout 900 python -c "
import sys, os
sys.path.insert(0, os.getcwd())
from nvdiffrecmc_amd import _build
_build.LIB = os.path.join(_build.BUILD, 'variants', 'libnvdr_hip.so.split16e2')
import pytest
sys.exit(pytest.main(['tests/test_gpu_bvh.py', 'tests/test_gpu_env_shade.py', '-q', '-m', 'gpu', '-x', '-k', 'visibility or invariants or 684k or vs_oracle or degenerate']))
" 2>&1 | tail -3
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -9 | tee $O/ab_bob1.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -9 | tee $O/ab_bob8.txt
PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -9 | tee $O/ab_dmtet800_1.txt
PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 AB_ONLY=split,split16e2 timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -7 | tee $O/ab_dmtet800_8.txt
PROBE_MESH=dmtet64_mid PROBE_VIEWS=1 PROBE_RES=800 AB_ONLY=split,split16e2 timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -7 | tee $O/ab_dmtet64_1.txt
