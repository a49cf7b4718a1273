#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s9
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -x -k "several_rank_schedule" 2>&1 | grep -E "^E |Error|assert" | head -40 | cut -c1-600 | tee gpurun_out/r5s9/forced.txt
