#!/bin/bash
# round 4, GPU session 31: G-buffer kernel with 8x8 tiles, debug arms compiled out, ADVICE fixes: affected suites + kernel times of the unlocked one-view iteration
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4s31; O=$R/gpurun_out/r4s31
timeout 1200 python -m pytest tests/test_gpu_gbuffer.py tests/test_gpu_geometry.py tests/test_gpu_env_shade.py tests/test_gpu_denoiser.py tests/test_gpu_bvh.py -q 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --config dmtet800 --batch 1 --graph off --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 20 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $O/kernel_trace_oneview_dmtet800_unlocked.md | cut -d'|' -f2-8,17,18 | cut -c1-150 | grep -i "gbuffer\|interpolate\|mesh_frame\|texture_lookup\|adam\|env_trace_kernel<false\|env_gen"
