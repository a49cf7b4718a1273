#!/bin/bash
# round 4, GPU session 40: three launches less per rebuild (counters / control words cleared by the hierarchy kernel, one-workgroup prefix sum for
# small trees): BVH suite, build-alone times, one-view bob
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s40; O=$R/gpurun_out/r4s40
timeout 900 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_gbuffer.py tests/test_gpu_geometry.py -q 2>&1 | tail -3
for sd in 3 2 0; do timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep triangles; done | tee $O/bvh_build_alone.txt
timeout 200 python tools/bvh_probe.py spot 0 40 2>&1 | grep triangles
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --steps 40 --warmup 10"
for rep in 1 2; do
timeout 300 python bench.py $COMMON --config bob512 --batch 1 --graph on > $O/bob1_$rep.json 2>$O/err.log
python - $O/bob1_$rep.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j['roofline']; f, b = r['forward_pass'], r['backward_pass']
print('bob1 ms/step %.3f (median %.3f)  fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f' % (j['ms_per_step'], j['median_ms_per_step'] or 0, f['gen_ms'], f['trace_ms'], f['shade_ms'], b['trace_ms'], b['shade_and_light_gradient_ms']))
PY
done
