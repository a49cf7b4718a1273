#!/bin/bash
# round 3, GPU session 36: image-loss backward without the target gradient; one-view iteration with the 8-row pair tiles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s36; O=gpurun_out/s36
timeout 600 python -m pytest tests/test_gpu_renderutils.py tests/test_gpu_denoiser.py -q -m gpu 2>&1 | tail -2
for g in on off; do
timeout 300 python bench.py --batch 1 --graph $g --no-cpu-baseline --no-pmc --no-large-mesh --steps 100 --warmup 20 2>/dev/null | tail -1 > $O/bench_oneview_graph_$g.json
python -c "import json; d=json.load(open('$O/bench_oneview_graph_$g.json')); print('one view graph $g', d['hip_graph'], d['ms_per_step'], d['median_ms_per_step'], d['steps_over_twice_the_median'])"
done
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-large-mesh --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_8.json
python -c "import json; d=json.load(open('$O/bench_8.json')); print('8 views', d['ms_per_step'], d['median_ms_per_step'], d['value'])"
