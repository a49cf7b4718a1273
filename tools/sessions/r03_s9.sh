#!/bin/bash
# round 3, GPU session 9: band-sorted light-gradient record blocks (parity + A/B against the previous build), compaction kernel,
# async build on/off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== gpu tests"; timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
echo "== A/B prev vs current (in process)"; AB_R2=0 PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | tail -14
echo "== A/B one view"; AB_R2=0 PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 4 2>&1 | tail -8
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 10 --warmup 3 > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | cut -c1-400
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_s9.md | cut -c1-60,100-190 | head -16
cd $R
for ab in 1 0 1 0; do
  echo "== NVDR_ASYNC_BUILD=$ab"; NVDR_ASYNC_BUILD=$ab timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-large-mesh --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['median_ms_per_step'], d['roofline']['forward_pass'])"
done
