#!/bin/bash
# round 4, GPU session 27: the driver's bench command with the new objects (other_configs, cpu_baseline_torch, exchange_world1, large_mesh unlocked)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s27; O=$R/gpurun_out/r4s27
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json
echo "bench wall seconds: $SECONDS"; tail -3 $O/bench_err.log
python - <<PY
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'median', d['median_ms_per_step'], 'it/s', d['iters_per_sec'])
print('roofline', d['roofline']['frac'], d['roofline']['kernel_ms_hip_events'], d['roofline'].get('hbm'), d['roofline'].get('l2'))
print('cpu', d.get('cpu_baseline'))
print('cpu_torch', d.get('cpu_baseline_torch'))
print('exchange', d['config'].get('exchange_world1'))
print('other', json.dumps(d.get('other_configs'), indent=1))
lm = d.get('large_mesh', {}); print('large', {k: lm.get(k) for k in ('ms_per_step','kernel_ms_hip_events','hbm','l2','geometry','error','seconds')})
print('params', d['config']['trained_parameters'], d['config']['parameter_bytes'])
PY
