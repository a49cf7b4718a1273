#!/bin/bash
# round 5, GPU session 6: count / emit kernels of the BVH collapse at 16 KB of LDS (128 threads, packed slots); whole suite; one-view and 8-view lines; build alone; one-view timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s6; O=$R/gpurun_out/r5s6
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest.txt
for sd in 0 2 3; do timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep triangles; done | tee $O/bvh_build_alone.txt
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view --steps 60 --warmup 10"
run() { name=$1; shift; timeout 300 python bench.py $COMMON "$@" > $O/$name.json 2>$O/$name.err; python - $O/$name.json $name <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j['roofline']; fw, bw = r['forward_pass'], r['backward_pass']
    print('%-14s ms/step %.3f graph %s | fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f | %s' % (sys.argv[2], j['ms_per_step'], j['hip_graph'], fw['gen_ms'], fw['trace_ms'], fw['shade_ms'], bw['trace_ms'], bw['shade_and_light_gradient_ms'],
          {k: v for k, v in (j['config'].get('exchange') or {}).items() if k in ('mode', 'policy', 'bytes_sent', 'tiles_touched', 'exposed_ms')}))
except Exception as e:
    print(sys.argv[2], 'FAILED', e, open(sys.argv[1].replace('.json', '.err')).read()[-800:])
PY
}
run bob1 --batch 1 --graph on
run bob1_x --batch 1 --graph on --exchange-world1
run bob1_xs --batch 1 --graph on --exchange-world1 --exchange sparse
run bob8
run dm8_locked --config dmtet800 --lock-pos on --steps 12 --warmup 4
run dm1_locked --config dmtet800 --batch 1 --graph on --lock-pos on
run dm1_trained --config dmtet800 --batch 1 --graph on
run dm1_trained_x --config dmtet800 --batch 1 --graph on --exchange-world1
cd /tmp; export TMPDIR=/tmp
trace() { name=$1; shift; rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py "$@" --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view > /tmp/kt.log 2>&1
  timeout 60 python $R/tools/rocpd_iteration.py /tmp/kt/r_results.db light_rows_kernel -8 > $O/${name}_iteration.txt 2>&1; head -1 $O/${name}_iteration.txt; }
trace kt_bob1 --batch 1 --graph on --steps 40 --warmup 10
