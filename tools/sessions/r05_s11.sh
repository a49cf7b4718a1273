#!/bin/bash
# round 5, GPU session 11: fused multiply-adds in the filter's tap; the driver's default bench command end to end (one_view, adam objects)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s11; O=$R/gpurun_out/r5s11
timeout 900 python -m pytest tests/test_gpu_denoiser.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py -q 2>&1 | grep -E "^E |passed|failed" | head -20 | cut -c1-300 | tee $O/pytest.txt
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench.json; echo "wall $SECONDS s"
python - <<'PY'
import json, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/r5s11'
try:
    d = json.load(open(O + '/bench.json'))
    print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['iters_per_sec'], d['config'].get('ms_per_step_cached_visibility'))
    r = d['roofline']; print('frac', r['frac'], r['kernel_ms_hip_events'], r.get('hbm', {}).get('hbm_frac'), r.get('l2'))
    print('stages', r['forward_pass'], r['backward_pass'])
    print('large', {k: v for k, v in d.get('large_mesh', {}).items() if k in ('ms_per_step', 'kernel_ms_hip_events', 'hbm', 'l2', 'node_steps_per_ray', 'seconds')})
    print('other', {k: (v.get('ms_per_step'), v.get('seconds'), v.get('trained_parameters')) for k, v in d.get('other_configs', {}).items()})
    print('adam', d['config'].get('adam'))
    print('one_view', json.dumps(d['config'].get('one_view'))[:1500])
    print('cpu', d['cpu_baseline']['value'], '| torch', d.get('cpu_baseline_torch', {}).get('value'))
except Exception as e:
    print('FAILED', e); print(open(O + '/bench_err.log').read()[-3000:])
PY
