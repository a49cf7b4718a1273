#!/bin/bash
# round 5, GPU session 18: confirmation at HEAD -- whole GPU suite, smoke, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s18; O=$R/gpurun_out/r5s18
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json; echo "wall $SECONDS s"
python - <<PY
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['iters_per_sec'], d['config'].get('ms_per_step_cached_visibility'))
r=d['roofline']; print(r['frac'], r['kernel_ms_hip_events'], r['hbm']['hbm_frac']); print('stages', r['forward_pass'], r['backward_pass'])
print('large_mesh', d['large_mesh'].get('ms_per_step'), d['large_mesh'].get('kernel_ms_hip_events'), d['large_mesh'].get('hbm', {}).get('hbm_frac'))
print('other', {k: v.get('ms_per_step') for k, v in d.get('other_configs', {}).items()})
ov = d['config'].get('one_view') or {}; print('one_view', {m: ov.get(m, {}).get('ms_per_step') for m in ('sparse','dense')}, (ov.get('projected_8gpu') or {}).get('dense'))
PY
