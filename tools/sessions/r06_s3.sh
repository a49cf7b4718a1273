#!/bin/bash
# round 6, GPU session 3: the split schedule with the rebuild on a HIGH-priority side stream; light-gradient gather fetching four blocks at a time (A/B against one)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s3; O=$R/gpurun_out/r6s3
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
bash tools/build_variants.sh lgnb1:"-DNVDR_LG_NB=1" lgnb2:"-DNVDR_LG_NB=2" lgnb8:"-DNVDR_LG_NB=8" 2>&1 | tail -3
PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 5 2>&1 | grep -v "Warning\|tenancy" | tee $O/ab_lg_bob1.txt
PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy" | tee $O/ab_lg_bob8.txt
PROBE_VIEWS=1 PROBE_RES=800 PROBE_SUBDIV=3 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -v "Warning\|tenancy" | tee $O/ab_lg_dmtet800_1.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_env_shade.py -q -m gpu -x -k "light_gradient or vs_oracle" 2>&1 | tail -3
