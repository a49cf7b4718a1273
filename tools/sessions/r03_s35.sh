#!/bin/bash
# round 3, GPU session 35: 8-row pair tiles as the default (p32 = the former 32 rows), 8-row tiles for the single-image kernels (s8); filter tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_denoiser.py -q -m gpu 2>&1 | tail -3
for v in 1 8; do echo "== views $v"; PROBE_VIEWS=$v timeout 300 python tools/dn_probe.py 5 2>&1 | tail -8; done
echo "== 800x800, 8 views"; PROBE_RES=800 PROBE_VIEWS=8 timeout 300 python tools/dn_probe.py 3 2>&1 | tail -8
