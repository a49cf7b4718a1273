#!/bin/bash
# round 6, GPU session 13: chunk stealing across the dealing counters (the XCDs finish a one-view launch at 47-82 % of its span): parity, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s13; O=$R/gpurun_out/r6s13
timeout 1500 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | grep -v Warning | tail -4 | tee $O/tests.txt
bash tools/build_variants.sh steal0:"-DNVDR_TRACE_STEALS=0" steal4:"-DNVDR_TRACE_STEALS=4" steal32:"-DNVDR_TRACE_STEALS=32" steal63:"-DNVDR_TRACE_STEALS=63" 2>&1 | tail -4
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -9 | tee $O/ab_bob1.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -9 | tee $O/ab_bob8.txt
PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -9 | tee $O/ab_dmtet800_1.txt
PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 AB_ONLY=steal0,steal32 timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -7 | tee $O/ab_dmtet800_8.txt
PROBE_MESH=dmtet64_mid PROBE_VIEWS=1 PROBE_RES=800 AB_ONLY=steal0,steal32 timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v "Warning\|tenancy\|amdgpu.ids" | head -7 | tee $O/ab_dmtet64_1.txt
PROBE_CASES="bob:512:0:1,bob:512:0:8" timeout 600 python tools/tail_probe.py $O/trace_phase_cycles.md 2>&1 | grep "^## \|mean end\|^timeline" | cut -c1-420
