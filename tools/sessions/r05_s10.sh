#!/bin/bash
# round 5, GPU session 10: the several-rank schedule with the rebuild joined at the end of the geometry stage; forced tests; one-view lines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s10; O=$R/gpurun_out/r5s10
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_exchange.py -q 2>&1 | grep -E "^E |passed|failed" | head -30 | cut -c1-400 | tee $O/pytest.txt
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --no-one-view --steps 60 --warmup 10"
for c in "--batch 1 --graph on" "--batch 1 --graph on --exchange-world1" "--batch 1 --graph on --exchange-world1 --exchange sparse" "--batch 1 --graph on --exchange-world1 --exchange dense" "--config dmtet800 --batch 1 --graph on --lock-pos on --exchange-world1" "--config dmtet800 --batch 1 --graph on --exchange-world1"; do
timeout 300 python bench.py $COMMON $c 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; fw,bw=r['forward_pass'],r['backward_pass']
print('$c', 'ms/step %.3f' % j['ms_per_step'], 'gen %.3f trace %.3f/%.3f shade %.3f/%.3f' % (fw['gen_ms'], fw['trace_ms'], bw['trace_ms'], fw['shade_ms'], bw['shade_and_light_gradient_ms']), {k: v for k, v in (j['config'].get('exchange') or {}).items() if k in ('mode','policy','bytes_sent','exposed_ms')})"
done 2>&1 | tee $O/lines.txt
