#!/bin/bash
# round 6, GPU session 9: confirmation after the stall fix (no memset nodes in captured paths), the G-buffer kernel's tile dealing, the bench with its extras in a child process
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s9; O=$R/gpurun_out/r6s9
echo "== gpu tests"; timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v Warning | tail -5 | tee $O/tests.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (driver's command)"; SECONDS=0; timeout 1200 python bench.py --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json; echo "wall $SECONDS s"
python - <<PY
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['iters_per_sec'], d['config'].get('ms_per_step_cached_visibility'))
r=d['roofline']; print(r['frac'], r['kernel_ms_hip_events'], r['valu']['active_lane_fraction'], r['hbm']['hbm_frac'])
print('flat', {k: v for k, v in d['config'].items() if 'one_view' in k or 'projected' in k})
for k in ('large_mesh', 'large_mesh_regular'):
    m = d.get(k) or {}
    print(k, {q: m.get(q) for q in ('ms_per_step', 'kernel_ms_hip_events', 'node_steps_per_ray', 'seconds', 'error')}, (m.get('hbm') or {}).get('hbm_frac'), (m.get('l2') or {}).get('l2_hit'))
print('validation', {q: (d.get('validation_n32') or {}).get(q) for q in ('ms_per_forward', 'trace_ms', 'rays_per_sec', 'seconds', 'error')})
print('other', {k: (v.get('ms_per_step'), v.get('seconds'), v.get('error')) for k, v in (d.get('other_configs') or {}).items()})
print('extras_note', d.get('extras_note'))
print('cpu', d['cpu_baseline']['value'], '| torch', (d.get('cpu_baseline_torch') or {}).get('value'))
PY
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['steps_over_twice_the_median'], {k: e.get(k) for k in ('mode','policy','bytes_sent','exposed_ms','geometry_stage_ms')} if e else '')" || tail -5 $O/$name.err; }
one oneview_bob512_graph_on --batch 1 --graph on
one oneview_dmtet800_unlocked_graph_on --config dmtet800 --batch 1 --graph on
one oneview_dmtet800_unlocked_schedule --config dmtet800 --batch 1 --graph on --exchange-world1
one oneview_dmtet64_800_unlocked_graph_on --config dmtet64_800 --batch 1 --graph on
one eightviews_dmtet800_unlocked --config dmtet800 --steps 20 --warmup 6
cd /tmp; export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py "$@" --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view --no-validation > /tmp/kt.log 2>&1
  timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/$name.md | cut -d'|' -f2-8 | cut -c1-150 | head -16
  timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel -8 > $O/${name}_iteration.txt 2>&1; head -1 $O/${name}_iteration.txt; }
trace kernel_trace_dmtet800_1view_unlocked --config dmtet800 --batch 1 --graph on --steps 30 --warmup 10
