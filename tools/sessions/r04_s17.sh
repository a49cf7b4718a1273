#!/bin/bash
# round 4, GPU session 17: whole GPU suite with the new BVH builder, then the 8-view / 1-view benches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s17; O=$R/gpurun_out/r4s17
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6
B="--no-cpu-baseline --no-pmc --no-large-mesh --steps 30 --warmup 10"
run() { name=$1; shift; timeout 400 python bench.py "$@" $B 2> $O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', 'ms/step %.3f median %.3f graph %s value %.3e' % (d['ms_per_step'], d['median_ms_per_step'], d['hip_graph'], d['value']))
except Exception as e: print('$name', 'FAILED', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run bob512 --config bob512
run bob512_1view --config bob512 --batch 1 --graph on
run dmtet800_unlocked --config dmtet800
run dmtet800_unlocked_1view --config dmtet800 --batch 1 --graph on
run dmtet800_locked --config dmtet800 --lock-pos on
run dmtet800_locked_1view --config dmtet800 --batch 1 --graph on --lock-pos on
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --config bob512 --batch 1 --graph off --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 10 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel | cut -c1-150 | tail -50
