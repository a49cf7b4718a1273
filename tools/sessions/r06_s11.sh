#!/bin/bash
# round 6, GPU session 11: the distributed tests after "split only with the dense exchange"; sparse one-view schedule (un-split again) repeated; bench default
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6s11; O=$R/gpurun_out/r6s11
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_gbuffer.py tests/test_gpu_geometry.py -q -m gpu -x 2>&1 | grep -v Warning | tail -4 | tee $O/tests.txt
one() { name=$1; shift; timeout 200 python bench.py "$@" --no-cpu-baseline --no-pmc --no-large-mesh --no-other-configs --no-one-view --no-validation --steps 100 --warmup 20 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json')); e=d['config'].get('exchange') or {}
print('$name', d['hip_graph'], round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), d['steps_over_twice_the_median'], {k: e.get(k) for k in ('mode','policy','bytes_sent','exposed_ms','geometry_stage_ms')} if e else '')" || tail -3 $O/$name.err | cut -c1-200; }
for i in 1 2 3; do one bob_sparse_$i --batch 1 --graph on --exchange-world1 --exchange sparse; done
for i in 1 2; do one bob_auto_$i --batch 1 --graph on --exchange-world1 --exchange auto; done
one bob_dense --batch 1 --graph on --exchange-world1
one dmtet800_locked_dense --config dmtet800 --lock-pos on --batch 1 --graph on --exchange-world1
one dmtet800_trained_dense --config dmtet800 --batch 1 --graph on --exchange-world1
echo "== bench (driver's command)"; SECONDS=0; timeout 1200 python bench.py --pmc-keep $O 2> $O/bench_err.log | tail -1 > $O/bench_bob512_n1.json; echo "wall $SECONDS s"
python - <<PY
import json; d=json.load(open('$O/bench_bob512_n1.json'))
print(d['value'], d['ms_per_step'], d['median_ms_per_step'])
print('flat', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d['config'].items() if 'one_view' in k and not isinstance(v, dict) or 'projected' in k})
print('sparse', d['config']['one_view'].get('sparse', {}).get('ms_per_step'), 'extras_note', d.get('extras_note'))
PY
