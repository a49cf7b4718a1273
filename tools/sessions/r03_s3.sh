#!/bin/bash
# round 3, GPU session 3: fma_mix micro-benchmark, parity of the mix variant, full suite, A/B, kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
B=nvdiffrecmc_amd/csrc/build
echo "== fma_mix ubench"; timeout 120 tools/ubench/fma_mix 2>&1 | tail -20
echo "== parity of the mix variant (bvh tests with the variant library in place)"
cp $B/libnvdr_hip.so /tmp/libnvdr_hip.so.orig; cp $B/variants/libnvdr_hip.so.mix $B/libnvdr_hip.so
timeout 600 python -m pytest tests/test_gpu_bvh.py -q -m gpu -x 2>&1 | tail -4
cp /tmp/libnvdr_hip.so.orig $B/libnvdr_hip.so
echo "== full suite"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8
echo "== stage probe (build time)"; PROBE_VIEWS=8 timeout 300 python tools/stage_probe.py 8,6,6 2>&1 | grep "bvh rebuild\|fwd gen\|production walk\|counting build"
echo "== ab bob 8 views"; PROBE_VIEWS=8 timeout 400 python tools/ab_inproc.py 4 2>&1 | grep -v tenancy | tail -12
echo "== ab dmtet800"; PROBE_VIEWS=8 PROBE_RES=800 PROBE_SUBDIV=3 AB_ITERS=3 timeout 600 python tools/ab_inproc.py 3 2>&1 | grep -v tenancy | tail -10
echo "== kernel trace"; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --steps 10 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_s3.md | cut -c1-130 | head -16
tail -1 /tmp/kt.log | cut -c1-300
