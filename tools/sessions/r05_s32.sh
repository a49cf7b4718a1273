#!/bin/bash
# round 5, GPU session 32: confirmation at the final HEAD (live-ray list in segments): the whole GPU suite, smoke, the driver's command with counters, the 8-view kernel trace,
# the one-view lines that DESIGN sections 6-7 quote
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/final2; O=$R/gpurun_out/final2
echo "== gpu tests"; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (driver's command)"; SECONDS=0; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json; echo "wall $SECONDS s"
python - <<PY
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['iters_per_sec'], d['config'].get('ms_per_step_cached_visibility'))
r=d['roofline']; print(r['frac'], r['kernel_ms_hip_events'], r['valu']['active_lane_fraction'], r['hbm']['hbm_frac'])
print('stages', {k: r['forward_pass'][k] for k in ('gen_ms','trace_ms','shade_ms')}, {k: r['backward_pass'][k] for k in ('trace_ms','shade_and_light_gradient_ms')})
lm=d.get('large_mesh', {}); print('large_mesh', lm.get('ms_per_step'), lm.get('kernel_ms_hip_events'), (lm.get('hbm') or {}).get('hbm_frac'))
print('other', {k: v.get('ms_per_step') for k, v in d.get('other_configs', {}).items()})
ov = d['config'].get('one_view') or {}
print('one_view', {m: ov.get(m, {}).get('ms_per_step') for m in ('sparse', 'dense')}, {m: v.get('speedup_vs_8_views_on_one_gpu') for m, v in (ov.get('projected_8gpu') or {}).items() if isinstance(v, dict)})
PY
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/kernel_trace_bob512_8views.md | cut -d'|' -f2-8 | cut -c1-150 | head -12
timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel -8 > $O/kernel_trace_bob512_8views_iteration.txt 2>&1; head -1 $O/kernel_trace_bob512_8views_iteration.txt
cd $R
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --steps 100 --warmup 20 2>/dev/null | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), round(d['config'].get('ms_per_step_cached_visibility') or 0,3), {k: e.get(k) for k in ('mode','bytes_sent','exposed_ms','geometry_stage_ms')} if e else '')"; }
one oneview_bob512_graph_on --batch 1 --graph on
one oneview_bob512_schedule_auto --batch 1 --graph on --exchange-world1
one oneview_dmtet800_unlocked_graph_on --config dmtet800 --batch 1 --graph on
one oneview_dmtet800_locked_graph_on --config dmtet800 --batch 1 --graph on --lock-pos on
one eightviews_dmtet800_locked --config dmtet800 --lock-pos on --steps 30 --warmup 6
