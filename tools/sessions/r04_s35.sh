#!/bin/bash
# round 4, GPU session 35: environment-only A/Bs -- generation kernel residency at one 800^2 view, LDS budget of the light-gradient gather
# (fewer bands = fewer placement rounds in the backward shading kernel), grid of the backward shading kernel at 256 spp, cost of the records
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s35; O=$R/gpurun_out/r4s35
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --steps 30 --warmup 8"
brief() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j['roofline']
    f, b = r['forward_pass'], r['backward_pass']
    print('%-30s ms/step %.3f (median %.3f)  fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f' % (sys.argv[2], j['ms_per_step'], j['median_ms_per_step'] or 0, f['gen_ms'], f['trace_ms'], f['shade_ms'], b['trace_ms'], b['shade_and_light_gradient_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
    print(open(sys.argv[1]).read()[-800:])
PY
}
run() {  # tag, env..., -- bench args
    local tag=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout 300 python bench.py $COMMON "$@" > $O/$tag.json 2> $O/$tag.err
    brief $O/$tag.json $tag
}
run dm1_gen4 NVDR_PBLOCKS=4,10,8 -- --config dmtet800 --batch 1 --graph on --lock-pos on
run dm1_gen3 NVDR_PBLOCKS=3,10,8 -- --config dmtet800 --batch 1 --graph on --lock-pos on
run dm1_default X=1 -- --config dmtet800 --batch 1 --graph on --lock-pos on
run bob8_lds96 NVDR_LG_LDS_KB=96 -- --config bob512
run bob8_lds128 NVDR_LG_LDS_KB=128 -- --config bob512
run bob8_lds160 NVDR_LG_LDS_KB=160 -- --config bob512
run bob8_lds64 NVDR_LG_LDS_KB=64 -- --config bob512
run bob8_norecords NVDR_DEBUG=2 -- --config bob512
run spot_b6 NVDR_PBLOCKS=10,6,6 -- --config spot512x256
run spot_b3 NVDR_PBLOCKS=10,6,3 -- --config spot512x256
run spot_b9 NVDR_PBLOCKS=10,6,9 -- --config spot512x256
run spot_b12 NVDR_PBLOCKS=10,6,12 -- --config spot512x256
