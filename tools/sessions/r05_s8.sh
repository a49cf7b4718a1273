#!/bin/bash
# round 5, GPU session 8: treelets of 32 leaves (two per wavefront) for meshes above 64 k triangles: suites, build alone, A/B on 684 k and 171 k triangles
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s8; O=$R/gpurun_out/r5s8
timeout 900 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_gbuffer.py tests/test_gpu_geometry.py tests/test_gpu_distributed.py -q 2>&1 | tail -8 | tee $O/pytest.txt
for sd in 0 2 3; do timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep triangles; done | tee $O/bvh_build_alone.txt
PROBE_VIEWS=8 PROBE_SUBDIV=3 PROBE_RES=800 AB_ITERS=3 timeout 900 python tools/ab_inproc.py 3 2>&1 | grep -v Warning | tee $O/ab_dmtet8.txt
PROBE_VIEWS=8 PROBE_SUBDIV=2 PROBE_RES=512 AB_ITERS=3 timeout 900 python tools/ab_inproc.py 3 2>&1 | grep -v Warning | tee $O/ab_171k.txt
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view --steps 40 --warmup 10"
for c in "--config dmtet800 --batch 1 --graph on --lock-pos on" "--config dmtet800 --batch 1 --graph on" "--config dmtet800 --batch 1 --graph on --exchange-world1"; do
timeout 300 python bench.py $COMMON $c 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; fw,bw=r['forward_pass'],r['backward_pass']
print('$c', 'ms/step %.3f' % j['ms_per_step'], 'gen %.3f trace %.3f/%.3f shade %.3f/%.3f' % (fw['gen_ms'], fw['trace_ms'], bw['trace_ms'], fw['shade_ms'], bw['shade_and_light_gradient_ms']), {k: v for k, v in (j['config'].get('exchange') or {}).items() if k in ('mode','policy','bytes_sent','exposed_ms')})"
done
