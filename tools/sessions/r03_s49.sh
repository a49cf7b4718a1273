#!/bin/bash
# round 3, GPU session 49: the driver's bench command at HEAD (the line behind profiles/r03_bench_bob512_n1.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/s49; O=gpurun_out/s49
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json
python -c "
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['iters_per_sec'], d['iters_per_sec_cached_visibility'])
print(d['roofline']['frac'], d['roofline']['kernel_ms_hip_events'])
print(d['large_mesh']['ms_per_step'], d['large_mesh']['kernel_ms_hip_events'], d['large_mesh']['hbm'])
"
