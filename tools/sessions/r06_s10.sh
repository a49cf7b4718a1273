#!/bin/bash
# round 6, GPU session 10: G-buffer kernel modes (strips / tiles x dealt / claimed); bob one view under the several-rank schedule, dense bucket, split vs unsplit
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s10; O=$R/gpurun_out/r6s10
for m in 0 1 2 3; do NVDR_TUNING=1 NVDR_GB_MODE=$m timeout 200 python tools/gb_probe.py 2>&1 | grep NVDR_GB; done | tee $O/gb_modes.txt
for m in 0 3; do NVDR_TUNING=1 NVDR_GB_MODE=$m PROBE_MESH=dmtet64_mid PROBE_SUBDIV=0 timeout 200 python tools/gb_probe.py 2>&1 | grep NVDR_GB; done | tee -a $O/gb_modes.txt
one() { name=$1; shift; timeout 300 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['steps_over_twice_the_median'], {k: e.get(k) for k in ('mode','policy','bytes_sent','exposed_ms','geometry_stage_ms')} if e else '')" || tail -5 $O/$name.err; }
one bob_dense_split --batch 1 --graph on --exchange-world1 --exchange dense
NVDR_TUNING=1 NVDR_SPLIT_STAGE2=0 one bob_dense_unsplit --batch 1 --graph on --exchange-world1 --exchange dense
one bob_dense_split_b --batch 1 --graph on --exchange-world1 --exchange dense
NVDR_TUNING=1 NVDR_SPLIT_STAGE2=0 one bob_dense_unsplit_b --batch 1 --graph on --exchange-world1 --exchange dense
one bob_auto_split --batch 1 --graph on --exchange-world1
one bob_plain --batch 1 --graph on
