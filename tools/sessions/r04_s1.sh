#!/bin/bash
# round 4, GPU session 1: the one-view numbers at HEAD nobody has: dmtet800 / bob512 with ONE view per
# iteration (the per-GPU share of an 8-GPU run), HIP graphs on, + kernel traces (graph off so that every kernel is a dispatch)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s1; O=$R/gpurun_out/r4s1
B="--no-cpu-baseline --no-pmc --no-large-mesh --steps 40 --warmup 10"
timeout 300 python bench.py --config dmtet800 --batch 1 --graph on $B 2> $O/e1.log | tail -1 > $O/oneview_dmtet800_graph_on.json
timeout 300 python bench.py --config bob512 --batch 1 --graph on $B 2> $O/e2.log | tail -1 > $O/oneview_bob512_graph_on.json
timeout 300 python bench.py --config dmtet800 $B 2> $O/e3.log | tail -1 > $O/dmtet800_8views.json
python - <<PY
import json
for f in ('oneview_dmtet800_graph_on','oneview_bob512_graph_on','dmtet800_8views'):
    try:
        d=json.load(open('$O/%s.json'%f)); print(f, d['ms_per_step'], d.get('median_ms_per_step'), d['hip_graph'], d['value'])
    except Exception as e: print(f, 'FAILED', e)
PY
cd /tmp; export TMPDIR=/tmp
for cfg in dmtet800 bob512; do
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --config $cfg --batch 1 --graph off --no-cpu-baseline --no-pmc --no-large-mesh --no-extended --steps 40 --warmup 10 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/kernel_trace_oneview_$cfg.md | cut -c1-200 | head -40
done
