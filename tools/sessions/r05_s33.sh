#!/bin/bash
# round 5, GPU session 33: the lines of DESIGN sections 6-7 that still dated from before the last kernel change, at the final HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/final3; O=$R/gpurun_out/final3
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --steps 100 --warmup 20 2>/dev/null | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), {k: e.get(k) for k in ('mode','bytes_sent','exposed_ms','geometry_stage_ms')} if e else '')"; }
one oneview_dmtet800_unlocked_schedule --config dmtet800 --batch 1 --graph on --exchange-world1
one oneview_dmtet800_locked_schedule --config dmtet800 --batch 1 --graph on --lock-pos on --exchange-world1
one oneview_bob512_schedule_sparse --batch 1 --graph on --exchange-world1 --exchange sparse
one eightviews_bob512_graph_on --graph on
full() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline --no-one-view --pmc-keep $O 2>/dev/null | tail -1 > $O/bench_$name.json
  python -c "import json; d=json.load(open('$O/bench_$name.json')); r=d['roofline']; print('$name', d['ms_per_step'], d['median_ms_per_step'], r['kernel_ms_hip_events'], r['frac'], r.get('hbm', {}).get('hbm_frac'), r.get('l2', {}).get('l2_hit'))"; }
full dmtet800_n1 --config dmtet800 --steps 10 --warmup 3
full spot512x256_n1 --config spot512x256 --steps 10 --warmup 3
full hotdog512x256_n1 --config hotdog512x256 --steps 8 --warmup 3
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --batch 1 --graph on --steps 40 --warmup 10 --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/kernel_trace_bob512_1view.md | cut -d'|' -f2-8 | cut -c1-150 | head -12
timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel -8 > $O/kernel_trace_bob512_1view_iteration.txt 2>&1; head -1 $O/kernel_trace_bob512_1view_iteration.txt
