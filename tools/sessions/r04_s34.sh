#!/bin/bash
# round 4, GPU session 34: hand-written radix sort in the LBVH build (csrc/sort_kernels.h) instead of rocPRIM: sort test, the BVH suite,
# build-alone times and the per-kernel table of the 684 k rebuild, one-view iterations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s34; O=$R/gpurun_out/r4s34
timeout 600 python -m pytest tests/test_gpu_bvh.py -q -x 2>&1 | tail -4
timeout 300 python -m pytest tests/test_gpu_gbuffer.py tests/test_gpu_geometry.py -q 2>&1 | tail -2
for sd in 3 2 0; do timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep triangles; done | tee $O/bvh_build_alone.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/tools/bvh_probe.py bob 3 40 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/bvh_build_684k_alone_kernel_trace.md | cut -d'|' -f2-8 | cut -c1-140 | head -24
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/tools/bvh_probe.py bob 0 40 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/bvh_build_bob_alone_kernel_trace.md | cut -d'|' -f2-8 | cut -c1-140 | head -24
cd $R
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --steps 40 --warmup 10"
for tag in bob1:"--config bob512 --batch 1 --graph on" dm1:"--config dmtet800 --batch 1 --graph on" bob8:"--config bob512"; do
  name=${tag%%:*}; args=${tag#*:}
  timeout 300 python bench.py $COMMON $args > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j['roofline']; f, b = r['forward_pass'], r['backward_pass']
    print('%-8s ms/step %.3f (median %.3f)  fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f' % (sys.argv[2], j['ms_per_step'], j['median_ms_per_step'] or 0, f['gen_ms'], f['trace_ms'], f['shade_ms'], b['trace_ms'], b['shade_and_light_gradient_ms']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1]).read()[-1500:])
PY
done
