#!/bin/bash
# round 3, GPU session 11: gather with compare-and-swap adds, merged record emission, Adam kernel, compaction
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== parity"; timeout 900 python -m pytest tests -q -m gpu -x -k "light or sparse or fullsize or env_shade or chunk or adam or renderutils" 2>&1 | tail -4
echo "== A/B prev / native atomics / current (in process)"; AB_R2=0 PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -B1 -A4 "^  current"
echo "== A/B one view"; AB_R2=0 PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -B1 -A4 "^  current"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 10 --warmup 3 > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | cut -c1-300
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_s11.md | cut -c1-60,100-190 | head -12
grep "compact_pixels\|light_grad\|adam" $R/gpurun_out/r03_kernel_trace_s11.md | cut -c1-60,100-190
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --batch 1 --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 20 --warmup 5 > /tmp/kt1.log 2>&1
tail -1 /tmp/kt1.log | cut -c1-300
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_oneview_s11.md | cut -c1-60,100-190 | head -12
grep "compact_pixels\|light_grad\|adam" $R/gpurun_out/r03_kernel_trace_oneview_s11.md | cut -c1-60,100-190
cd $R
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-large-mesh --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8 views', d['ms_per_step'], d['median_ms_per_step'], d['value'])"
timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-pmc --no-large-mesh --steps 50 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1 view', d['hip_graph'], d['ms_per_step'], d['median_ms_per_step'])"
