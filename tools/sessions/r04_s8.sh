#!/bin/bash
# round 4, GPU session 8: all GPU tests with the reference's parameter set in the harness + first benches of it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s8; O=$R/gpurun_out/r4s8
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
B="--no-cpu-baseline --no-pmc --no-large-mesh --steps 30 --warmup 10"
run() { name=$1; shift; timeout 400 python bench.py "$@" $B 2> $O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', 'ms/step %.3f median %.3f graph %s value %.3e params %s MB' % (d['ms_per_step'], d['median_ms_per_step'], d['hip_graph'], d['value'], d['config']['parameter_bytes']/1e6))
except Exception as e: print('$name', 'FAILED', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run bob512_full --config bob512
run bob512_r3 --config bob512 --material-set r3
run bob512_full_1view --config bob512 --batch 1 --graph on
run bob512_r3_1view --config bob512 --batch 1 --graph on --material-set r3
run dmtet800_full_unlocked --config dmtet800
run dmtet800_full_locked --config dmtet800 --lock-pos on
run dmtet800_full_unlocked_1view --config dmtet800 --batch 1 --graph on
run dmtet800_full_locked_1view --config dmtet800 --batch 1 --graph on --lock-pos on
