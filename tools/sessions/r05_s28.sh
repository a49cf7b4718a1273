#!/bin/bash
# round 5, GPU session 28: shadow-ray kernel with the top of the tree in LDS (workgroups of 1024 threads): visibility parity (brute force, all meshes), env-shade parity,
# then A/B inside one process: one workgroup per CU with a 1472-node image (product candidate) | the round-4 launch shape (NVDR_TRACE_TOP=0) | two workgroups, 256 nodes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s28; O=$R/gpurun_out/r5s28
timeout 900 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_env_shade.py -q -x 2>&1 | grep -v Warning | tail -6 | tee $O/pytest.txt
AB_ENV="top0:NVDR_TRACE_TOP=0|top2:NVDR_TRACE_TOP=2" PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
AB_ENV="top0:NVDR_TRACE_TOP=0|top2:NVDR_TRACE_TOP=2" PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
