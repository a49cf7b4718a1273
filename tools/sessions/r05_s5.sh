#!/bin/bash
# round 5, GPU session 5: treelet rebuild at the bottom of the Morton tree (bvh.hip): BVH / G-buffer / geometry suites, build times alone,
# in-process A/B against the plain Karras tree and against larger-area-first child order, on bob (8 views) and on 684 k triangles; the forced-schedule test with its error
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s5; O=$R/gpurun_out/r5s5
timeout 900 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_gbuffer.py tests/test_gpu_geometry.py tests/test_gpu_env_shade.py -q 2>&1 | tail -15 | tee $O/pytest_bvh.txt
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -x -k "several_rank_schedule" 2>&1 | grep -v "^$" | tail -60 | cut -c1-400 | tee $O/pytest_forced.txt
for sd in 0 2 3; do timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep triangles; done | tee $O/bvh_build_alone.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 AB_ITERS=3 timeout 900 python tools/ab_inproc.py 3 2>&1 | grep -v Warning | tee $O/ab_dmtet8.txt
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
