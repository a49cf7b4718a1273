#!/bin/bash
# round 3, GPU session 10: LDS atomic rates; gather with every workgroup walking all bands; filter tile sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== LDS atomics ubench"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-atomic-alignment tools/ubench/lds_atomic.hip -o /tmp/lds_atomic 2>/dev/null && timeout 120 /tmp/lds_atomic | tee gpurun_out/r03_lds_atomic_ubench.txt
echo "== light-gradient parity"; timeout 900 python -m pytest tests -q -m gpu -x -k "light or sparse or fullsize or env_shade or chunk" 2>&1 | tail -4
echo "== filter variants, 8 views"; AB_ONLY=by8,by32 PROBE_VIEWS=8 timeout 300 python tools/dn_probe.py 5 2>&1 | tail -5
echo "== filter variants, 1 view"; AB_ONLY=by8,by32 PROBE_VIEWS=1 timeout 300 python tools/dn_probe.py 5 2>&1 | tail -5
echo "== A/B prev vs current (in process)"; AB_ONLY=prev AB_R2=0 PROBE_VIEWS=8 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -A3 "gen  "
echo "== A/B one view"; AB_ONLY=prev AB_R2=0 PROBE_VIEWS=1 timeout 600 python tools/ab_inproc.py 4 2>&1 | grep -A3 "gen  "
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 10 --warmup 3 > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | cut -c1-300
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_s10.md | cut -c1-60,100-190 | head -14
grep "compact_pixels\|light_grad_reduce" $R/gpurun_out/r03_kernel_trace_s10.md | cut -c1-60,100-190
