#!/bin/bash
# round 3: the measurement set behind DESIGN.md / profiles/r03_* (driver's bench command, counters, kernel traces, one-view graph on/off)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/final
O=gpurun_out/final
if [ -z "$SKIP_TESTS" ]; then
echo "== gpu tests"; timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
echo "== bench (driver's command)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json
python -c "
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['iters_per_sec'], d['iters_per_sec_cached_visibility'])
print(d['roofline']['frac'], d['roofline']['kernel_ms_hip_events'], d['roofline']['valu']['active_lane_fraction'], d['roofline']['hbm']['hbm_frac'], d['roofline']['l2'])
print(d.get('large_mesh'))
print(d['cpu_baseline']['value'])
"
ls $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 20 --warmup 5 > /tmp/kt.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/$O/kernel_trace.md | cut -c1-60,100-190 | head -14
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --batch 1 --no-cpu-baseline --no-pmc --no-extended --no-large-mesh --steps 40 --warmup 10 > /tmp/kt1.log 2>&1
timeout 60 python $R/tools/rocpd_summary.py /tmp/kt/r_results.db $R/$O/kernel_trace_oneview.md | cut -c1-60,100-190 | head -12
cd $R
for g in on off; do
timeout 300 python bench.py --batch 1 --graph $g --no-cpu-baseline --no-pmc --no-large-mesh --steps 100 --warmup 20 2>/dev/null | tail -1 > $O/bench_oneview_graph_$g.json
python -c "import json; d=json.load(open('$O/bench_oneview_graph_$g.json')); print('one view graph $g', d['hip_graph'], d['ms_per_step'], d['median_ms_per_step'], d['steps_over_twice_the_median'])"
done
if [ -n "$MORE" ]; then
timeout 600 python bench.py --config dmtet800 --no-cpu-baseline --steps 10 --warmup 3 --pmc-keep $O 2>/dev/null | tail -1 > $O/bench_dmtet800_n1.json
python -c "import json; d=json.load(open('$O/bench_dmtet800_n1.json')); print('dmtet800', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['kernel_ms_hip_events'], d['roofline']['hbm'])"
timeout 600 python bench.py --config spot512x256 --no-cpu-baseline --steps 10 --warmup 3 --pmc-keep $O 2>/dev/null | tail -1 > $O/bench_spot512x256_n1.json
python -c "import json; d=json.load(open('$O/bench_spot512x256_n1.json')); print('spot512x256', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['kernel_ms_hip_events'], d['roofline']['frac'])"
timeout 600 python bench.py --config hotdog512x256 --no-cpu-baseline --steps 8 --warmup 3 --pmc-keep $O 2>/dev/null | tail -1 > $O/bench_hotdog512x256_n1.json
python -c "import json; d=json.load(open('$O/bench_hotdog512x256_n1.json')); print('hotdog512x256', d['ms_per_step'], d['median_ms_per_step'], d['roofline']['kernel_ms_hip_events'], d['roofline']['frac'], d['roofline']['hbm'])"
fi
