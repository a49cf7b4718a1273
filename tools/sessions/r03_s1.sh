#!/bin/bash
# round 3, GPU session 1: new BVH8 kernel correctness, then A/B against the round-2 kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== bvh tests"; timeout 600 python -m pytest tests/test_gpu_bvh.py -q -m gpu -x -s 2>&1 | tail -25
echo "== full suite"; timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
echo "== stage probe bob 8 views"; PROBE_VIEWS=8 timeout 300 python tools/stage_probe.py 8,6,6 2>&1 | grep -v "^wave\|host side"
echo "== ab bob 8 views"; PROBE_VIEWS=8 timeout 300 python tools/ab_inproc.py 4 2>&1 | tail -12
echo "== ab bob 1 view"; PROBE_VIEWS=1 timeout 300 python tools/ab_inproc.py 4 2>&1 | tail -12
echo "== ab dmtet800"; PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 AB_ITERS=3 timeout 600 python tools/ab_inproc.py 3 2>&1 | tail -12
