#!/bin/bash
# round 3, GPU session 22: HIP graphs at 8 views; kernel trace after the pair filter
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for g in on off on off; do
timeout 300 python bench.py --graph $g --no-cpu-baseline --no-pmc --no-large-mesh --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph $g: 8 views', d['hip_graph'], d['ms_per_step'], d['median_ms_per_step'])"
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 20 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_s22.md | cut -c1-60,100-190 | head -30
