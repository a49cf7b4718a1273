#!/bin/bash
# round 4, GPU session 37: after factoring the per-sample arithmetic (shade_sample) and the record placement (RecordBlocks) out of the three
# stage-3 kernels: parity suites, then the iteration times of bob512 (8 views, 1 view) and spot512x256
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s37; O=$R/gpurun_out/r4s37
timeout 1200 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py tests/test_gpu_renderutils.py -q 2>&1 | tail -3
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --steps 30 --warmup 8"
for tag in bob8:"--config bob512" bob1:"--config bob512 --batch 1 --graph on" spot:"--config spot512x256" bob8b:"--config bob512"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 300 python bench.py $COMMON $args > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j['roofline']; f, b = r['forward_pass'], r['backward_pass']
    print('%-8s ms/step %.3f (median %.3f)  fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f' % (sys.argv[2], j['ms_per_step'], j['median_ms_per_step'] or 0, f['gen_ms'], f['trace_ms'], f['shade_ms'], b['trace_ms'], b['shade_and_light_gradient_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1]).read()[-1500:])
PY
done
