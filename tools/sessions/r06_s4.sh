#!/bin/bash
# round 6, GPU session 4: the decorrelated-seed test's failure; refit policy (tests + one-view numbers of the trained 684 k mesh); phase-clock builds of the traversal kernel;
# per-kernel traces of the one-view iterations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s4; O=$R/gpurun_out/r6s4
timeout 600 python -m pytest tests/test_gpu_env_shade.py -q -m gpu -x -k "decorrelated" --tb=short 2>&1 | grep -v Warning | tail -30 | tee $O/test_decorrelated.txt
timeout 1500 python -m pytest tests/test_gpu_bvh.py tests/test_gpu_geometry.py tests/test_gpu_distributed.py -q -m gpu -x -k "refit or two_rank or several_rank" 2>&1 | grep -v Warning | tail -12 | tee $O/tests_refit.txt
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['steps_over_twice_the_median'], {k: e.get(k) for k in ('mode','policy','bytes_sent','exposed_ms','geometry_stage_ms')} if e else '')" || tail -5 $O/$name.err; }
one oneview_dmtet800_unlocked_graph_on_k8 --config dmtet800 --batch 1 --graph on
one oneview_dmtet800_unlocked_graph_on_k1 --config dmtet800 --batch 1 --graph on --rebuild-every 1
one oneview_dmtet800_unlocked_schedule_k8 --config dmtet800 --batch 1 --graph on --exchange-world1
one oneview_dmtet800_unlocked_schedule_k1 --config dmtet800 --batch 1 --graph on --exchange-world1 --rebuild-every 1
one eightviews_dmtet800_unlocked_k8 --config dmtet800 --steps 20 --warmup 6
one eightviews_dmtet800_unlocked_k1 --config dmtet800 --steps 20 --warmup 6 --rebuild-every 1
timeout 600 python tools/tail_probe.py $O/trace_phase_cycles.md 2>&1 | grep -v Warning | tail -70
cd /tmp; export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py "$@" --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view --no-validation > /tmp/kt.log 2>&1
  timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/$name.md | cut -d'|' -f2-8 | cut -c1-150 | head -24
  timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel -8 > $O/${name}_iteration.txt 2>&1; head -1 $O/${name}_iteration.txt; }
trace kernel_trace_bob512_1view --batch 1 --graph on --steps 40 --warmup 10
trace kernel_trace_bob512_1view_schedule --batch 1 --graph on --steps 40 --warmup 10 --exchange-world1
trace kernel_trace_dmtet800_1view_unlocked --config dmtet800 --batch 1 --graph on --steps 30 --warmup 10
