#!/bin/bash
# round 4, GPU session 33: pixel-local queue shading kernels for S > 64 (env_shade_local_kernel): parity, then the 256-spp presets with the plain
# kernels (NVDR_SHADE_QUEUE=0) and the new ones; one-view generation kernel with 3 instead of 4 resident workgroups per CU (room for the build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s33; O=$R/gpurun_out/r4s33
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "local_queue or config3 or config5 or queue_shading" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_env_shade.py -q 2>&1 | tail -3
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --steps 20 --warmup 5"
brief() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j['roofline']
    f, b = r['forward_pass'], r['backward_pass']
    print('%-44s ms/step %.3f (median %.3f)  fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f' % (sys.argv[2], j['ms_per_step'], j['median_ms_per_step'] or 0, f['gen_ms'], f['trace_ms'], f['shade_ms'], b['trace_ms'], b['shade_and_light_gradient_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
    print(open(sys.argv[1]).read()[-1500:])
PY
}
run() {  # tag, env..., -- bench args
    local tag=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout 300 python bench.py $COMMON "$@" > $O/$tag.json 2> $O/$tag.err
    brief $O/$tag.json $tag
}
for rep in 1 2; do
run spot_plain_$rep NVDR_SHADE_QUEUE=0 -- --config spot512x256
run spot_local_$rep NVDR_SHADE_QUEUE=3 -- --config spot512x256
done
run hotdog_plain NVDR_SHADE_QUEUE=0 -- --config hotdog512x256
run hotdog_local NVDR_SHADE_QUEUE=3 -- --config hotdog512x256
run bob1_gen4_a NVDR_PBLOCKS=4,10,3 -- --config bob512 --batch 1 --graph on --steps 40 --warmup 10
run bob1_gen3_a NVDR_PBLOCKS=3,10,3 -- --config bob512 --batch 1 --graph on --steps 40 --warmup 10
run bob1_gen4_b NVDR_PBLOCKS=4,10,3 -- --config bob512 --batch 1 --graph on --steps 40 --warmup 10
run bob1_gen3_b NVDR_PBLOCKS=3,10,3 -- --config bob512 --batch 1 --graph on --steps 40 --warmup 10
run bob1_gen2 NVDR_PBLOCKS=2,10,3 -- --config bob512 --batch 1 --graph on --steps 40 --warmup 10
