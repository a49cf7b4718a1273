#!/bin/bash
# round 5, GPU session 4: fused input copy, output zero-fill folded into the pixel compaction, deferred BVH launches (build mode 2), pack by kernel:
# the suite, then one-view timings in HIP graphs for build modes 1 / 2, the several-rank schedule at world 1, and a timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s4; O=$R/gpurun_out/r5s4
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view --steps 60 --warmup 10"
run() { name=$1; shift; timeout 300 python bench.py $COMMON "$@" > $O/$name.json 2>$O/$name.err; python - $O/$name.json $name <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j['roofline']; fw, bw = r['forward_pass'], r['backward_pass']
    print('%-14s ms/step %.3f graph %s | fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f | %s' % (sys.argv[2], j['ms_per_step'], j['hip_graph'], fw['gen_ms'], fw['trace_ms'], fw['shade_ms'], bw['trace_ms'], bw['shade_and_light_gradient_ms'],
          {k: v for k, v in (j['config'].get('exchange') or {}).items() if k in ('mode', 'bytes_sent', 'tiles_touched', 'exposed_ms')}))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
export NVDR_TUNING=1
for rep in 1 2; do
NVDR_BUILD_MODE=1 run bob1_m1_$rep --batch 1 --graph on
NVDR_BUILD_MODE=2 run bob1_m2_$rep --batch 1 --graph on
done
NVDR_BUILD_MODE=1 run bob1_xs_m1 --batch 1 --graph on --exchange-world1
NVDR_BUILD_MODE=2 run bob1_xs_m2 --batch 1 --graph on --exchange-world1
NVDR_BUILD_MODE=2 run bob1_xd_m2 --batch 1 --graph on --exchange-world1 --exchange dense
unset NVDR_BUILD_MODE
run bob8 
run dm1_locked --config dmtet800 --batch 1 --graph on --lock-pos on
run dm1_trained --config dmtet800 --batch 1 --graph on
run dm1_trained_x --config dmtet800 --batch 1 --graph on --exchange-world1
cd /tmp; export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py "$@" --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view > /tmp/kt.log 2>&1
  timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel -8 > $O/${name}_iteration.txt 2>&1; head -1 $O/${name}_iteration.txt; }
NVDR_BUILD_MODE=2 trace kt_bob1_m2 --batch 1 --graph on --steps 40 --warmup 10
NVDR_BUILD_MODE=2 trace kt_bob1_xs_m2 --batch 1 --graph on --steps 40 --warmup 10 --exchange-world1
