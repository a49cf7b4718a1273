#!/bin/bash
# round 6, GPU session 25: confirmation at the final HEAD: the whole GPU suite, smoke(), the driver's bench command (with its counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s25; O=$R/gpurun_out/r6s25
echo "== gpu tests"; timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench"; SECONDS=0; timeout 1200 python bench.py --gpus 1 --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json; echo "wall $SECONDS s"
python - <<PY
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print(d['value'], d['ms_per_step'], d['median_ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_hip_events'])
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d['config'].items() if ('one_view' in k or 'projected' in k) and not isinstance(v, dict)})
PY
