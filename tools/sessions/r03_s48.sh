#!/bin/bash
# round 3, GPU session 48: BVH fit kernel with write-through hand-off instead of per-node fences: BVH tests, dmtet800 / hotdog / bob iteration
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s48; O=gpurun_out/s48
timeout 600 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_gbuffer.py -q -m gpu 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "dmtet or config5 or training_step" 2>&1 | tail -2
for cfg in dmtet800 hotdog512x256 bob512; do
timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-pmc --no-large-mesh --no-extended --steps 8 --warmup 3 2>/dev/null | tail -1 > $O/$cfg.json
python -c "import json; d=json.load(open('$O/$cfg.json')); r=d['roofline']; print('$cfg', d['ms_per_step'], r['forward_pass']['gen_ms'], r['kernel_ms_hip_events'])"
done
