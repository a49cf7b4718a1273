#!/bin/bash
# round 6, GPU session 22: A/B of the refill's list entries fetched one refill ahead (NVDR_TRACE_PRELIVE), then the round's measurement set at this HEAD (tools/r06_final.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s22; O=$R/gpurun_out/r6s22
bash tools/build_variants.sh prelive:"-DNVDR_TRACE_PRELIVE=1" 2>&1 | tail -1
timeout 900 python -c "
import sys, os
sys.path.insert(0, os.getcwd())
from nvdiffrecmc_amd import _build
_build.LIB = os.path.join(_build.BUILD, 'variants', 'libnvdr_hip.so.prelive')
import pytest
sys.exit(pytest.main(['tests/test_gpu_bvh.py', 'tests/test_gpu_env_shade.py', '-q', '-m', 'gpu', '-x', '-k', 'visibility or 684k or vs_oracle or degenerate or odd_list or split_walk']))
" 2>&1 | tail -2
ab() { out=$1; shift; env "$@" timeout 900 python tools/ab_inproc.py ${ROUNDS:-5} 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -5 | tee $O/$out.txt; }
ab ab_bob8 PROBE_VIEWS=8
ab ab_bob1 PROBE_VIEWS=1
ab ab_684k_1 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3
ab ab_dmtet64_1 PROBE_MESH=dmtet64_mid PROBE_VIEWS=1 PROBE_RES=800
rm -f nvdiffrecmc_amd/csrc/build/variants/*
bash tools/r06_final.sh
