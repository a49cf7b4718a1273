#!/bin/bash
# round 5, GPU session 1: the whole GPU suite after the exchange / trainer rework, then the one-view schedule (world-1 RCCL) and a short default line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r5s2; O=$R/gpurun_out/r5s2
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/pytest.txt
COMMON="--no-cpu-baseline --no-pmc --no-extended --no-large-mesh --no-other-configs --steps 40 --warmup 10"
timeout 300 python bench.py $COMMON --no-one-view --config bob512 > $O/bob8.json 2>$O/bob8.err
timeout 300 python bench.py $COMMON --no-one-view --config bob512 --batch 1 --graph on > $O/bob1.json 2>$O/bob1.err
timeout 300 python bench.py $COMMON --no-one-view --config bob512 --batch 1 --graph on --exchange-world1 > $O/bob1_x.json 2>$O/bob1_x.err
timeout 300 python bench.py $COMMON --no-one-view --config bob512 --batch 1 --graph on --exchange-world1 --exchange dense > $O/bob1_xd.json 2>$O/bob1_xd.err
python - <<'PY'
import json, glob, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/r5s2'
for f in sorted(glob.glob(O + '/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'no line', e, open(f.replace('.json', '.err')).read()[-1500:]); continue
    r = j['roofline']; fw, bw = r['forward_pass'], r['backward_pass']
    print('%-12s ms/step %.3f graph %s | fwd gen %.3f trace %.3f shade %.3f | bwd trace %.3f shade %.3f | exchange %s' % (
        os.path.basename(f), j['ms_per_step'], j['hip_graph'], fw['gen_ms'], fw['trace_ms'], fw['shade_ms'], bw['trace_ms'], bw['shade_and_light_gradient_ms'],
        json.dumps(j['config'].get('exchange'))))
PY
