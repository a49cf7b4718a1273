#!/bin/bash
# round 6, GPU session 15: the split-walk build as a kernel of its own, chosen by the context's last launch: whole GPU suite, one-view lines, default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s15; O=$R/gpurun_out/r6s15
echo "== gpu tests"; timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -4 | tee $O/tests.txt
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}; r=d['roofline']
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), 'trace fwd/bwd', round(r['forward_pass']['trace_ms'],3), round(r['backward_pass']['trace_ms'],3), {k: e.get(k) for k in ('mode','exposed_ms','geometry_stage_ms')} if e else '')" || tail -3 $O/$name.err | cut -c1-200; }
one bob_1view --batch 1 --graph on
NVDR_TUNING=1 NVDR_TRACE_SPLIT_MODE=0 one bob_1view_plain_build --batch 1 --graph on
one bob_1view_schedule --batch 1 --graph on --exchange-world1
one dmtet800_1view_trained --config dmtet800 --batch 1 --graph on
NVDR_TUNING=1 NVDR_TRACE_SPLIT_MODE=0 one dmtet800_1view_trained_plain_build --config dmtet800 --batch 1 --graph on
one dmtet800_1view_trained_schedule --config dmtet800 --batch 1 --graph on --exchange-world1
one dmtet800_1view_locked_schedule --config dmtet800 --lock-pos on --batch 1 --graph on --exchange-world1
one bob_8views --steps 50 --warmup 10
NVDR_TUNING=1 NVDR_TRACE_SPLIT_MODE=0 one bob_8views_plain_build --steps 50 --warmup 10
