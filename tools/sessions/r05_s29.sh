#!/bin/bash
# round 5, GPU session 29: the LDS image of the tree's top at EIGHT wavefronts per SIMD (session 28's variants ran at four: 66 VGPRs): two workgroups of 1024 threads with 192
# nodes each (NVDR_TRACE_TOP=2), eight of 256 threads with 48 nodes each (=8), against the round-4 launch shape (=0) and one workgroup with 1472 nodes (=1, current default)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s29; O=$R/gpurun_out/r5s29
NVDR_TRACE_TOP=8 timeout 600 python -m pytest tests/test_gpu_bvh.py -q -x 2>&1 | grep -v Warning | tail -3 | tee $O/pytest_top8.txt
AB_ENV="top0:NVDR_TRACE_TOP=0|top2:NVDR_TRACE_TOP=2|top8:NVDR_TRACE_TOP=8" PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
