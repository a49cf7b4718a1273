#!/bin/bash
# round 4, GPU session 9: geometry-unlocked dmtet800 (8 views / 1 view) with a sane position learning rate + kernel trace of the one-view case
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s9; O=$R/gpurun_out/r4s9
B="--no-cpu-baseline --no-pmc --no-large-mesh --steps 30 --warmup 10"
run() { name=$1; shift; timeout 400 python bench.py "$@" $B 2> $O/$name.err | tail -1 > $O/$name.json; python - <<PY
import json
try:
    d=json.load(open('$O/$name.json')); print('$name', 'ms/step %.3f median %.3f graph %s value %.3e params %s MB' % (d['ms_per_step'], d['median_ms_per_step'], d['hip_graph'], d['value'], d['config']['parameter_bytes']/1e6))
except Exception as e: print('$name', 'FAILED', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run dmtet800_full_unlocked --config dmtet800
run dmtet800_full_unlocked_1view --config dmtet800 --batch 1 --graph on
run hotdog_unlocked --config hotdog512x256
cd /tmp; export TMPDIR=/tmp
for cfg in dmtet800 bob512; do
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --config $cfg --batch 1 --graph off --no-cpu-baseline --no-pmc --no-large-mesh --no-extended --steps 40 --warmup 10 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/kernel_trace_oneview_$cfg.md | cut -d'|' -f2-8,17,18 | cut -c1-200 | head -64
done
