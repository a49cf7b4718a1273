#!/bin/bash
# round 3, GPU session 34: pair-filter tile height (32 rows: current, x*; 8 rows: q*) x LDS read pipelining depth (0, 1, 2 = current, 4), one view and eight
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for v in 1 8; do echo "== views $v"; PROBE_VIEWS=$v timeout 300 python tools/dn_probe.py 5 2>&1 | tail -20; done
