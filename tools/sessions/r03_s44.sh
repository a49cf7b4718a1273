#!/bin/bash
# round 3, GPU session 44: queue kernels only where a pixel is one round; env-shade + full-size suites; spot preset check
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s44
timeout 900 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --config spot512x256 --no-cpu-baseline --no-pmc --steps 8 --warmup 3 2>/dev/null | tail -1 > gpurun_out/s44/spot.json
python -c "import json; d=json.load(open('gpurun_out/s44/spot.json')); print('spot512x256', d['ms_per_step'], d['median_ms_per_step'])"
