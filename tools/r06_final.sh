#!/bin/bash
# round 6: the measurement set behind DESIGN.md / profiles/r06_* at the final HEAD (driver's bench command with counters and its child-process extras, kernel traces of
# the 8-view and the one-view iterations, one-view numbers in HIP graphs -- plain and under the several-rank schedule with a one-rank RCCL group -- for the locked and the
# geometry-trained presets, the marching-tets presets, the BVH build alone)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/final
O=$R/gpurun_out/final
if [ -z "$SKIP_TESTS" ]; then
echo "== gpu tests"; timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
echo "== bench (driver's command)"; SECONDS=0; timeout 1200 python bench.py --gpus 1 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json; echo "wall $SECONDS s"
python - <<PY
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['iters_per_sec'], d['config'].get('ms_per_step_cached_visibility'))
r=d['roofline']; print(r['frac'], r['kernel_ms_hip_events'], r['valu']['active_lane_fraction'], r['hbm']['hbm_frac'], r['l2'])
print('stages', r['forward_pass'], r['backward_pass'])
print('flat', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d['config'].items() if ('one_view' in k or 'projected' in k) and not isinstance(v, dict)})
for k in ('large_mesh', 'large_mesh_regular'):
    m = d.get(k) or {}
    print(k, {q: m.get(q) for q in ('ms_per_step', 'kernel_ms_hip_events', 'kernel_rays_per_sec', 'node_steps_per_ray', 'seconds', 'error')}, (m.get('hbm') or {}).get('hbm_frac'), (m.get('l2') or {}).get('l2_hit'))
print('validation', {q: (d.get('validation_n32') or {}).get(q) for q in ('ms_per_forward', 'trace_ms', 'rays_per_sec', 'seconds', 'error')})
print('other', {k: (v.get('ms_per_step'), v.get('seconds'), v.get('error')) for k, v in (d.get('other_configs') or {}).items()})
print('extras_note', d.get('extras_note'), '| cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], '| torch', (d.get('cpu_baseline_torch') or {}).get('value'), '| adam', d['config'].get('adam'))
PY
cd /tmp; export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py "$@" --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view --no-validation > /tmp/kt.log 2>&1
  timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/$name.md | cut -d'|' -f2-8 | cut -c1-150 | head -14
  timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel -8 > $O/${name}_iteration.txt 2>&1; head -1 $O/${name}_iteration.txt; }
trace kernel_trace_bob512_8views --steps 20 --warmup 5
trace kernel_trace_bob512_1view --batch 1 --graph on --steps 40 --warmup 10
trace kernel_trace_dmtet800_1view_unlocked --config dmtet800 --batch 1 --graph on --steps 30 --warmup 10
trace kernel_trace_dmtet64_800_8views --config dmtet64_800 --steps 6 --warmup 3
cd $R
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json   # (stderr kept: one of these lines came back empty once, session 29 could not reproduce it)
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['steps_over_twice_the_median'], {k: e.get(k) for k in ('mode','policy','bytes_sent','exposed_ms','geometry_stage_ms')} if e else '')"; }
one oneview_bob512_graph_on --batch 1 --graph on
one oneview_bob512_schedule_dense --batch 1 --graph on --exchange-world1
one oneview_bob512_schedule_sparse --batch 1 --graph on --exchange-world1 --exchange sparse
one oneview_dmtet800_unlocked_graph_on --config dmtet800 --batch 1 --graph on
one oneview_dmtet800_unlocked_schedule --config dmtet800 --batch 1 --graph on --exchange-world1
one oneview_dmtet800_unlocked_graph_on_rebuild_every_1 --config dmtet800 --batch 1 --graph on --rebuild-every 1
one oneview_dmtet800_locked_graph_on --config dmtet800 --batch 1 --graph on --lock-pos on
one oneview_dmtet800_locked_schedule --config dmtet800 --batch 1 --graph on --lock-pos on --exchange-world1
one oneview_dmtet64_800_unlocked_graph_on --config dmtet64_800 --batch 1 --graph on
one eightviews_dmtet800_unlocked --config dmtet800 --steps 30 --warmup 6
one eightviews_dmtet800_locked --config dmtet800 --lock-pos on --steps 30 --warmup 6
one eightviews_bob512_graph_on --graph on
full() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-one-view --pmc-keep $O 2>/dev/null | tail -1 > $O/bench_$name.json
  python -c "import json; d=json.load(open('$O/bench_$name.json')); r=d['roofline']; print('$name', d['ms_per_step'], d['median_ms_per_step'], r['kernel_ms_hip_events'], r['frac'], r.get('hbm', {}).get('hbm_frac'), r.get('l2', {}).get('l2_hit'))"; }
full dmtet64_800_n1 --config dmtet64_800 --steps 8 --warmup 3
full dmtet800_n1 --config dmtet800 --steps 10 --warmup 3
full spot512x256_n1 --config spot512x256 --steps 10 --warmup 3
full hotdog512x256_n1 --config hotdog512x256 --steps 8 --warmup 3
full dmtet64_512x256_n1 --config dmtet64_512x256 --steps 4 --warmup 2
for sd in 3 2 0; do timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep triangles; done | tee $O/bvh_build_alone.txt
for m in dmtet64_mid dmtet64_init; do timeout 200 python tools/bvh_probe.py $m 0 40 2>&1 | grep triangles; done | tee -a $O/bvh_build_alone.txt
PROBE_CASES="bob:512:0:1,bob:512:0:8,bob:800:3:1,dmtet64_mid:800:0:1" timeout 600 python tools/tail_probe.py $O/trace_phase_cycles.md > /dev/null 2>&1; grep -c "^## " $O/trace_phase_cycles.md
