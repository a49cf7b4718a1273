#!/bin/bash
# round 5, GPU session 17: two-rank training with trained geometry under the pipelined schedule; the dmtet800 locked schedule line again (process-group teardown fix)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s17; O=$R/gpurun_out/r5s17
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -k "two_rank_training" 2>&1 | grep -E "^E |passed|failed|Error" | head -30 | cut -c1-400 | tee $O/pytest.txt
timeout 300 python bench.py --config dmtet800 --batch 1 --graph on --lock-pos on --exchange-world1 --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --steps 100 --warmup 20 2>/dev/null | tail -1 > $O/oneview_dmtet800_locked_schedule.json
python -c "
import json; d=json.load(open('$O/oneview_dmtet800_locked_schedule.json')); e=d['config'].get('exchange') or {}
print('dm1 locked schedule', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), {k: e.get(k) for k in ('mode','policy','bytes_sent','exposed_ms','geometry_stage_ms')})"
