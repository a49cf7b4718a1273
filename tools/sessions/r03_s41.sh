#!/bin/bash
# round 3, GPU session 41: queue shading kernels as the default -- whole GPU suite, bench (8 views, one view)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s41; O=gpurun_out/s41
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-large-mesh --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_8.json
python -c "import json; d=json.load(open('$O/bench_8.json')); print('8 views', d['ms_per_step'], d['median_ms_per_step'], d['value'], d['iters_per_sec_cached_visibility'])"
for g in on; do
timeout 300 python bench.py --batch 1 --graph $g --no-cpu-baseline --no-pmc --no-large-mesh --steps 100 --warmup 20 2>/dev/null | tail -1 > $O/bench_oneview_graph_$g.json
python -c "import json; d=json.load(open('$O/bench_oneview_graph_$g.json')); print('one view graph $g', d['hip_graph'], d['ms_per_step'], d['median_ms_per_step'], d['steps_over_twice_the_median'])"
done
