#!/bin/bash
# round-2 GPU session B (short): tests with concise output, gather after the chunk fix, queue variants A/B, slow-mode-per-context probe, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
echo "== tests"
timeout 1300 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids" | tail -60 | cut -c1-220 | tee $O/r02b_tests.txt
echo "== stage probe 8 views: gather | atomics | one-chunk budget"
PROBE_VIEWS=8 timeout 200 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen" | tee $O/r02b_stage.txt
NVDR_DEBUG=16 PROBE_VIEWS=8 timeout 200 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen" | tee -a $O/r02b_stage.txt
NVDR_STREAM_BUDGET_MB=8192 PROBE_VIEWS=8 timeout 200 python tools/stage_probe.py 8,6,6 2>&1 | grep -E "fwd gen" | tee -a $O/r02b_stage.txt
echo "== queue variants (8 views)"
PROBE_VIEWS=8 AB_WITH_CURRENT=1 bash tools/ab_run.sh 2>&1 | grep -E "==|fwd gen" | tee $O/r02b_ab8.txt
echo "== slow mode per context"
for i in 1 2 3; do timeout 120 python tools/mode_ctx_probe.py 5 2>&1 | grep "one process"; done | tee $O/r02b_modectx.txt
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r02b_bench.json 2> $O/r02b_bench.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02b_bench.json').read().strip().splitlines()[-1])
r=j['roofline']
print({k:j[k] for k in ('value','ms_per_step','median_ms_per_step','iters_per_sec','iters_per_sec_cached_visibility','n_gpus')})
print({k:r.get(k) for k in ('bound','achieved','peak','frac','traffic','kernel_ms_hip_events')})
print('valu',r.get('valu')); print('hbm',r.get('hbm')); print('l2',r.get('l2')); print('fwd',r['forward_pass']); print('bwd',r['backward_pass']); print(r.get('pmc_note'))
print('cpu', j['cpu_baseline'])
PY
tail -3 $O/r02b_bench.err | cut -c1-300
