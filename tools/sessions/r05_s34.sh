#!/bin/bash
# round 5, GPU session 34: 8 instead of 16 segments of the live-ray list (variant `segs8`), in-process A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s34; O=$R/gpurun_out/r5s34
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 7 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 7 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
