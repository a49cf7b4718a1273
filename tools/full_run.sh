#!/bin/bash
# full GPU validation of the current build: tests, smoke, stage probe, bench (with cpu baseline), kernel trace summary
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/stage_probe.py 6,6,4 2>&1 | grep "fwd gen"
timeout 600 python bench.py ${BENCH_ARGS} 2>&1 | tail -1 | tee gpurun_out/bench_full.json
if [ -z "$NO_TRACE" ]; then
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/kernel_trace.md | cut -c1-150 | head -16
fi
