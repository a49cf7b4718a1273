#!/bin/bash
# round 4, GPU session 36: traversal launch shape at ONE view (5-13 M rays): resident workgroups per CU (8 / 6 / 4), refill threshold (16 / 8 / 32),
# coarse chunks -- in-process interleaved A/B (tools/ab_inproc.py), bob 512^2 and the 684 k-triangle mesh at 800^2; gbuffer backward without
# materialised zero gradients (geometry tests)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s36; O=$R/gpurun_out/r4s36
timeout 300 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_renderutils.py -q 2>&1 | tail -2
AB_R2=0 PROBE_VIEWS=1 timeout 300 python tools/ab_inproc.py 7 2>&1 | grep -v "^\[nvdr\]" | tee $O/ab_bob_1view.txt | head -12
AB_R2=0 PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3 timeout 400 python tools/ab_inproc.py 5 2>&1 | grep -v "^\[nvdr\]" | tee $O/ab_dmtet_1view.txt | head -12
AB_R2=0 PROBE_VIEWS=8 AB_ONLY=rf8,rf32 timeout 300 python tools/ab_inproc.py 5 2>&1 | grep -v "^\[nvdr\]" | tee $O/ab_bob_8views.txt | head -8
