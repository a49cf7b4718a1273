#!/bin/bash
# round 3, GPU session 2: full suite, bench with PMC, refill-threshold variants, kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== full suite"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 --pmc-keep gpurun_out 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/r03_bench_bob512_s2.json; python tools/bench_brief.py gpurun_out/r03_bench_bob512_s2.json 2>&1 | head -40
echo "== ab bob 8 views"; PROBE_VIEWS=8 timeout 400 python tools/ab_inproc.py 4 2>&1 | grep -v tenancy | tail -16
echo "== kernel trace"; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --steps 10 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_s2.md | cut -c1-160 | head -24
