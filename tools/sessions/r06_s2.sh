#!/bin/bash
# round 6, GPU session 2: (1) the split several-rank schedule (locked geometry: rebuild as a graph of its own on a side stream, stage 2 cut in front of the traversal):
# the two-rank / one-rank-schedule tests, then the one-view numbers; (2) per-phase cycles + launch timeline of the traversal kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s2; O=$R/gpurun_out/r6s2
timeout 1200 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_env_shade.py -q -m gpu -x 2>&1 | grep -v Warning | tail -8 | tee $O/tests.txt
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['steps_over_twice_the_median'], {k: e.get(k) for k in ('mode','policy','bytes_sent','exposed_ms','geometry_stage_ms')} if e else '')" || tail -5 $O/$name.err; }
one oneview_bob512_graph_on --batch 1 --graph on
one oneview_bob512_schedule_auto --batch 1 --graph on --exchange-world1
NVDR_TUNING=1 NVDR_SPLIT_STAGE2=0 one oneview_bob512_schedule_auto_unsplit --batch 1 --graph on --exchange-world1
one oneview_dmtet800_locked_graph_on --config dmtet800 --batch 1 --graph on --lock-pos on
one oneview_dmtet800_locked_schedule --config dmtet800 --batch 1 --graph on --lock-pos on --exchange-world1
NVDR_TUNING=1 NVDR_SPLIT_STAGE2=0 one oneview_dmtet800_locked_schedule_unsplit --config dmtet800 --batch 1 --graph on --lock-pos on --exchange-world1
one oneview_dmtet800_unlocked_graph_on --config dmtet800 --batch 1 --graph on
one oneview_dmtet800_unlocked_schedule --config dmtet800 --batch 1 --graph on --exchange-world1
timeout 600 python tools/tail_probe.py $O/trace_phase_cycles.md 2>&1 | grep -v Warning | tail -80
