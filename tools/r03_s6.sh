#!/bin/bash
# round 3, GPU session 6: gather (batched loads) + two-row denoiser: tests, kernel traces (8 views, 1 view), bench with the large_mesh object
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8
echo "== kernel trace 8 views"; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 10 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/gpurun_out/r03_kernel_trace_s6.md > /dev/null; cut -c1-60,100-175 $R/gpurun_out/r03_kernel_trace_s6.md | head -24
echo "== kernel trace 1 view"; rm -rf /tmp/kt1
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt1 -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --batch 1 --graph off --steps 20 --warmup 5 > /tmp/kt1.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt1/r_results.db $R/gpurun_out/r03_kernel_trace_oneview_s6.md > /dev/null; cut -c1-60,100-175 $R/gpurun_out/r03_kernel_trace_oneview_s6.md | head -30
tail -1 /tmp/kt1.log | cut -c1-400
cd $R
echo "== bench (with large_mesh)"; timeout 900 python bench.py --steps 20 --warmup 5 --pmc-keep gpurun_out 2>gpurun_out/bench_err.log | tail -1 > gpurun_out/r03_bench_bob512_s6.json; python tools/bench_brief.py gpurun_out/r03_bench_bob512_s6.json 2>&1 | cut -c1-600 | head -12; python -c "
import json; j=json.loads(open('gpurun_out/r03_bench_bob512_s6.json').read().strip().splitlines()[-1]); print(json.dumps(j.get('large_mesh'))[:1500])"
echo "== one view graph on/off"; for g in off on; do timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-large-mesh --batch 1 --graph $g --steps 50 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r03_oneview_graph_$g.json; python -c "
import json; j=json.loads(open('gpurun_out/r03_oneview_graph_$g.json').read()); print('$g', {k: j[k] for k in ('ms_per_step','median_ms_per_step','min_ms_per_step','max_ms_per_step','steps_over_twice_the_median','hip_graph')})"; done
echo "== PMC gather kernel"; cd $R
bash tools/pmc_run.sh r03_gather light_grad_band -- python $R/bench.py --pmc-child --config bob512 --steps 2 --warmup 1 2>&1 | cut -c1-190 | tail -40
