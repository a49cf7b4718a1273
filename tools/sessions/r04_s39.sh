#!/bin/bash
# round 4, GPU session 39: covered pixels listed in 32 x 32 tiles instead of raster order (locality of the shadow-ray origins inside the traversal's
# moving window): parity suites, then A/B NVDR_PIX_TILE=0/1 on the 684 k mesh, the 171 k mesh, bob
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s39; O=$R/gpurun_out/r4s39
timeout 1200 python -m pytest tests/test_gpu_env_shade.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs"
brief() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j['roofline']; f, b = r['forward_pass'], r['backward_pass']
    print('%-14s ms/step %.3f (median %.3f)  fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f' % (sys.argv[2], j['ms_per_step'], j['median_ms_per_step'] or 0, f['gen_ms'], f['trace_ms'], f['shade_ms'], b['trace_ms'], b['shade_and_light_gradient_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1]).read()[-1500:])
PY
}
for t in 0 1 0 1; do
NVDR_PIX_TILE=$t timeout 300 python bench.py $COMMON --config dmtet800 --lock-pos on --steps 10 --warmup 4 > $O/dm8_$t.json 2>$O/err.log; brief $O/dm8_$t.json dm8_tile$t
done
for t in 0 1; do
NVDR_PIX_TILE=$t timeout 300 python bench.py $COMMON --config hotdog512x256 --lock-pos on --steps 8 --warmup 3 > $O/hd_$t.json 2>$O/err.log; brief $O/hd_$t.json hotdog_tile$t
NVDR_PIX_TILE=$t timeout 300 python bench.py $COMMON --config bob512 --steps 30 --warmup 8 > $O/bob8_$t.json 2>$O/err.log; brief $O/bob8_$t.json bob8_tile$t
NVDR_PIX_TILE=$t timeout 300 python bench.py $COMMON --config dmtet800 --lock-pos on --batch 1 --graph on --steps 30 --warmup 8 > $O/dm1_$t.json 2>$O/err.log; brief $O/dm1_$t.json dm1_tile$t
done
