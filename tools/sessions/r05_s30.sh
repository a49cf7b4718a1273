#!/bin/bash
# round 5, GPU session 30: the generation kernel after its set-up rework -- is the single claim counter of the live list (109 k same-address atomics at ~12 ns) what bounds it now?
# timing variant with the claims spread over 64 counters (the list it writes is garbage: only the generation kernel's own time means anything)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s30; O=$R/gpurun_out/r5s30
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob8.txt
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v Warning | tee $O/ab_bob1.txt
