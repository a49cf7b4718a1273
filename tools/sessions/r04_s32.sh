#!/bin/bash
# round 4, GPU session 32: hoisted re-trace (nvdr_ctx_set_retrace_ahead): parity tests, then A/B of the iteration time with the walk inside the
# backward launch (NVDR_RETRACE_AHEAD=0) and hoisted with 4 / 6 / 8 resident workgroups per CU, one view and eight views
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s32; O=$R/gpurun_out/r4s32
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "hoisted or config4_684k or training_step or batched_views or fused_and_composed" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_distributed.py -q 2>&1 | tail -3
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --steps 40 --warmup 10"
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
run bob1_inside_$rep NVDR_RETRACE_AHEAD=0 -- --config bob512 --batch 1 --graph on
run bob1_hoist4_$rep NVDR_RETRACE_BLOCKS=4 -- --config bob512 --batch 1 --graph on
run bob1_hoist6_$rep NVDR_RETRACE_BLOCKS=6 -- --config bob512 --batch 1 --graph on
run bob1_hoist8_$rep NVDR_RETRACE_BLOCKS=8 -- --config bob512 --batch 1 --graph on
done
run bob8_inside NVDR_RETRACE_AHEAD=0 -- --config bob512
run bob8_hoist4 NVDR_RETRACE_BLOCKS=4 -- --config bob512
run bob8_hoist6 NVDR_RETRACE_BLOCKS=6 -- --config bob512
run bob8_hoist8 NVDR_RETRACE_BLOCKS=8 -- --config bob512
run dm1_inside NVDR_RETRACE_AHEAD=0 -- --config dmtet800 --batch 1 --lock-pos on --graph on
run dm1_hoist6 NVDR_RETRACE_BLOCKS=6 -- --config dmtet800 --batch 1 --lock-pos on --graph on
run dm1_hoist4 NVDR_RETRACE_BLOCKS=4 -- --config dmtet800 --batch 1 --lock-pos on --graph on
