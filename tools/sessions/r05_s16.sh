#!/bin/bash
# round 5, GPU session 16: is the generation kernel held up by its live-list claims (one atomic counter, ~97 k claims per 8-view launch)?  timing variant with the claims spread over 64 counters
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s16; O=$R/gpurun_out/r5s16
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
