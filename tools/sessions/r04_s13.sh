#!/bin/bash
# round 4, GPU session 13: BVH build alone (684 k / 171 k / 10.7 k triangles): event times and the per-kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s13; O=$R/gpurun_out/r4s13
timeout 900 python -m pytest tests/test_gpu_bvh.py -x -q 2>&1 | tail -3
for sd in 3 2 0; do timeout 200 python tools/bvh_probe.py bob $sd 40 2>&1 | grep triangles; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/tools/bvh_probe.py bob 3 40 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/kernel_trace_bvh_build_684k_alone.md | cut -d'|' -f2-8,17,18 | cut -c1-170 | grep -i "bvh\|rocprim\|kernel |\|fill\|copy"
